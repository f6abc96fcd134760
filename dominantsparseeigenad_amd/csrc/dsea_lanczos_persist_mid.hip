// dsea_lanczos_persist_mid.hip -- the k-step Lanczos loop of reference Lanczos.py:49-77 as ONE launch for MID-SIZE
// halo-1 operators: the 3-point stencil of reference examples/schrodinger1D.py:18-27 at 8 192 < n <= 131 072 rows
// (BASELINE configs[2]: N = 100 000, k = 300).
//
// Why a second single-launch form.  dsea_lanczos_persist.hip covers n <= 8192 with <= 64 workgroups that each gather ALL
// partial sums.  At N = 1e5 the multi-launch loop spends ~22 of its 47 us per step on the fixed latencies of four
// dependent launches, and the rest streaming a basis (240 MB of fp64 at k = 300) that does not fit any cache -- but a
// third of it fits the chip's REGISTERS and LDS: 256 CUs x (512 KB of VGPRs + 160 KB of LDS) = 172 MB.  Here every one
// of the G <= 256 workgroups (one per CU, 8 waves) owns a slab of R = ceil(n / G) rows for the whole run and keeps
//   * its rows of the first NL basis vectors in LDS,
//   * its rows of the next 4 x NR vectors in the registers of its 4 waves (one wave per SIMD: 256 VGPRs + 256 AGPRs
//     each -- a kernel without MFMA leaves the AGPR half to the register allocator, which parks the slots there),
//   * and streams only the vectors beyond those from HBM: fp64 for the dots pass, the bf16 storage shadow of the basis
//     (same premise max|c_j| <= tau ||r||, checked on the device every step, fp64 stream otherwise) for
//     the correction pass.
// At N = 1e5, k = 300: 40 + 40 vectors on chip, i.e. the first 80 steps touch no HBM at all and the streamed traffic of
// the whole run is 54 % of the multi-launch form's.
//
// Per step three grid-wide exchanges through data-tagged granules (dsea_device.h; cdna_hip_programming.md Guideline 16,
// form R2: relaxed agent-scope stores / polls, epoch = step + 1, buffers zeroed per launch, bounded spins):
//
//   X1   every workgroup publishes {||r||^2, r.A0 r, r_first, r_last} of its slab (A0 = the stencil with zero halos);
//        everybody gathers all G records: beta = ||r||, alpha = r.A r / ||r||^2 (the slab-boundary cross terms
//        2 coef r_last(g) r_first(g+1) are added from the edge rows -- the mat-vec is applied to the UN-normalised r, by
//        linearity, which is what lets the two scalar reductions of Lanczos.py:69-72 travel together, exactly as the
//        row-partitioned driver does), and the two halo rows of the own slab.
//        q_s = r / beta (stored), u = A r / beta, r' = u - alpha q_s - beta q_{s-1}               Lanczos.py:61,69-72
//   X2a  partial c_j = q_j . r' (j <= s) and ||r'||^2 go to the OWNER of j (workgroup j mod G), which sums the G
//        partials in a fixed order: a direct all-to-all of (s + 1) x G partials would have every CU read 1.2 MB
//   X2b  the owners publish the reduced coefficients, everybody gathers them;  r = r' - sum_j c_j q_j   Lanczos.py:66
//
// Every workgroup sums the same values in the same order: alpha, beta, c are bit-identical everywhere, so is the
// breakdown decision (device-side record as in the multi-launch form) and the premise decision.  Runs are bit-repeatable.
// T agrees with the multi-launch form to rounding (different summation order; alpha = r.Ar/||r||^2 instead of q.Aq).
// A lost peer (device shared with other work) makes the bounded spins give up: fail flag -> dsea_lanczos_status returns
// DSEA_ERR_TIMEOUT and the host repeats the run on the multi-launch kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dsea_internal.h"
#include "dsea_device.h"

namespace dsea {

namespace {
typedef gran_u64 lzm_gu64;
#define LZM_TIMEOUT_TICKS DSEA_GRANULE_TIMEOUT_TICKS
#ifndef LZM_WAVES
#define LZM_WAVES 4               /* one wave per SIMD: each may use the full 512 registers (256 VGPR + 256 AGPR) */
#endif
#define LZM_THREADS (64 * LZM_WAVES)
#define LZM_MAX_K 512
#define LZM_MAX_G 256
#define LZM_NCH 4                 /* 128-row chunks of a slab held per lane: R <= 512 rows */
#ifndef LZM_NR
#define LZM_NR 10                 /* register-cache slots per wave (8 doubles per lane each).  The wave is alone on its SIMD:
                                     256 VGPRs + 256 AGPRs, and the compiler parks the slots in the AGPR half on its own (no
                                     MFMA here).  10 is what fits without scratch (resource-usage: 256 + 237).  Tried and
                                     measured with -Rpass-analysis=kernel-resource-usage: explicit v_accvgpr_write / _read
                                     residency ("a" constraints; volatile writes so that the conditional slot update is not
                                     if-converted into VGPR selects; tied "+a" operands; an unconditional shift-register
                                     form) -- the allocator keeps TWO registers per loop-carried dword in every form, i.e.
                                     32 row-pair slots fill the 256 AGPRs: no more capacity than the compiler finds itself.
                                     A wave-partitioned layout (each wave a quarter of the slab, 124-VGPR loop body, no
                                     cross-wave combine) was built and measured as well: 17.8 instead of 19.4 us per step with
                                     everything on chip, but its quarter-row streams (784-byte fp64 and 196-byte bf16 loads
                                     per wave) ran slower than this form's: 46.5 vs 43.5 us per step at N = 1e5, k = 300. */
#endif
//                  /* register-cache slots per wave (one slot = this wave's copy of a slab of one vector) */
#define LZM_LDS_BYTES 163840      /* 160 KiB per CU */

__device__ __forceinline__ double lzm_stencil_row(double coef, double Vi, double xi, double up, double dn) {
  const double lap = __dadd_rn(__dadd_rn(__dmul_rn(-2.0, xi), up), dn);
  return __dadd_rn(__dmul_rn(coef, lap), __dmul_rn(Vi, xi));
}

// one lane's records of the workgroups lane, lane + 64, lane + 128, lane + 192 (NF doubles each): poll until every one
// carries `epoch`; absent workgroups (>= G) read as zeros
template <int NF>
__device__ __forceinline__ bool lzm_poll_records(lzm_gu64* buf, int G, int lane, unsigned epoch, double (&v)[4][NF],
                                                 long long t0) {
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int gg = lane + 64 * m;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        v[m][f] = 0.0;
        if (gg < G) ok &= granule_try_get(buf + ((int64_t)gg * NF + f) * 2, epoch, v[m][f]);
      }
    }
    if (ok) return true;
    __builtin_amdgcn_s_sleep(1);
    if (wall_clock64() - t0 > LZM_TIMEOUT_TICKS) return false;
  }
}
}  // namespace

struct LzmArgs {
  Stencil3Params st;
  const double* q0;
  double* Q;
  int64_t ldq, n;
  int k;
  uint16_t* Qs;     // bf16 shadow of the basis (k rows x lds) or null
  int64_t lds;
  double tau;       // premise bound of the shadow pass
  double* alphas;
  double* betas;
  double* brk;      // [0] breakdown step, [1] running scale
  double* fail;     // set to 1 when a peer did not arrive in time
  double* lp_count; // [0] steps whose streamed correction read the shadow (or streamed nothing), [1] fp64 fallbacks
  unsigned long long* comm;
  int G, R, kslots, NL;
  int lose_peer;    // test hook: the last workgroup exits at once (its peers must time out, not hang)
};

// dynamic LDS carve (doubles), all offsets even (16-byte aligned): see lzm_lds_doubles()
//   s_c[K2] s_cpart[K2] s_b[32] s_r[R+4] s_y[R] s_V[R] s_q[2][R] s_part[8][R] s_cache[NL][R]   (K2 = LZM_MAX_K + 2, even)
#define LZM_K2 (LZM_MAX_K + 2)

__global__ __launch_bounds__(LZM_THREADS, LZM_WAVES / 4) void k_lanczos_persist_mid(LzmArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lzm_smem[];
  const int R = a.R;
  double* s_c = lzm_smem;
  double* s_cpart = s_c + LZM_K2;
  double* s_b = s_cpart + LZM_K2;       // [1] fail [2] left halo [3] right halo [5] premise violated [8 + 4 m ..] X1 wave sums [24 + m] r_last of workgroup 64 m + 63
  double* s_r = s_b + 32;               // s_r[2 + local row] (pairs 16-byte aligned); [1] and [2 + R] are zero halos
  double* s_y = s_r + (R + 4);
  double* s_V = s_y + R;
  double* s_q = s_V + R;                // [2][R]: q_s at parity s & 1 (q_{s-1} at the other): not held in registers
  double* s_part = s_q + 2 * R;         // [wave][R]
  double* s_cache = s_part + LZM_WAVES * R;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int g = blockIdx.x, G = a.G;
  if (a.lose_peer && G > 1 && g == G - 1) return;
  const int64_t n = a.n, base = (int64_t)g * R;
  const int Rg = (int)((n - base < R) ? (n - base) : R);          // rows of this slab (>= 1 by construction)
  const int NL = a.NL, M = NL + LZM_WAVES * LZM_NR;               // vectors j < M live on chip
  lzm_gu64* X1 = (lzm_gu64*)a.comm;                               // [G][4]
  lzm_gu64* X2a = X1 + (int64_t)G * 4 * 2;                        // [kslots][G]
  lzm_gu64* X2b = X2a + (int64_t)a.kslots * G * 2;                // [kslots]
  const double coef = a.st.coef;

  // lane-local rows: chunk c -> local rows lr = 128 c + 2 lane, lr + 1
  bool vx[LZM_NCH], vy[LZM_NCH];
#pragma unroll
  for (int c = 0; c < LZM_NCH; ++c) {
    const int lr = 128 * c + 2 * lane;
    vx[c] = lr < Rg;
    vy[c] = lr + 1 < Rg;
  }
  double2 r[LZM_NCH];
  double2 qc[LZM_NR > 0 ? LZM_NR : 1][LZM_NCH];                   // register cache: vectors NL + WAVES * slot + wave
#pragma unroll
  for (int c = 0; c < LZM_NCH; ++c) {
    const int lr = 128 * c + 2 * lane;
    r[c] = make_double2(0.0, 0.0);
    if (vx[c]) r[c].x = a.q0[base + lr];
    if (vy[c]) r[c].y = a.q0[base + lr + 1];
  }
#pragma unroll
  for (int sl = 0; sl < LZM_NR; ++sl)
#pragma unroll
    for (int c = 0; c < LZM_NCH; ++c) qc[sl][c] = make_double2(0.0, 0.0);
  for (int t = tid; t < R; t += LZM_THREADS) s_V[t] = (t < Rg) ? a.st.V[base + t] : 0.0;
  for (int t = tid; t < 2 * R; t += LZM_THREADS) s_q[t] = 0.0;
  if (tid == 0) {
    s_b[1] = 0.0;
    s_b[5] = 0.0;
    s_r[0] = s_r[1] = 0.0;
    s_r[R + 2] = s_r[R + 3] = 0.0;
  }
  double scale = 0.0;                           // running max |alpha|, |beta| (same in every thread of every workgroup)
  int lp_steps = 0, fb_steps = 0;
  __syncthreads();

#ifdef DSEA_LZM_TIMING
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = wall_clock64();
#define LZM_TICK(k) { const long long tn = wall_clock64(); tacc[k] += tn - tprev; tprev = tn; }
#else
#define LZM_TICK(k)
#endif
  const int lane_fixed = lane;
  for (int s = 0; s < a.k; ++s) {
    const unsigned epoch = (unsigned)(s + 1);
    const long long t0 = wall_clock64();
    // an opaque copy of the lane index per step: without it the compiler hoists the ~100 loop-invariant per-lane addresses
    // and predicates of this body out of the step loop and keeps them alive beside the register cache
    int lane = lane_fixed;
    asm volatile("" : "+v"(lane));
    // ---------------------------------------------------------------- X1: ||r||^2, r.A0 r, edge rows
    if (wv == 0) {
#pragma unroll
      for (int c = 0; c < LZM_NCH; ++c) {
        const int lr = 128 * c + 2 * lane;
        if (lr < R) *reinterpret_cast<double2*>(s_r + 2 + lr) = r[c];   // (rows >= Rg hold zeros: the zero halo of a short slab)
      }
      // wave 0 alone writes s_r here and reads it below: its own LDS operations complete in order
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
      double prr = 0.0, pra = 0.0;
#pragma unroll
      for (int c = 0; c < LZM_NCH; ++c) {
        const int lr = 128 * c + 2 * lane;
        double2 y = make_double2(0.0, 0.0);
        if (lr < R) {
          const double dn = s_r[lr + 1], up = s_r[lr + 4];      // rows lr - 1 and lr + 2
          const double2 Vv = *reinterpret_cast<const double2*>(s_V + lr);
          if (vx[c]) y.x = lzm_stencil_row(coef, Vv.x, r[c].x, r[c].y, dn);
          if (vy[c]) y.y = lzm_stencil_row(coef, Vv.y, r[c].y, up, r[c].x);
          *reinterpret_cast<double2*>(s_y + lr) = y;
        }
        prr = fma(r[c].x, r[c].x, fma(r[c].y, r[c].y, prr));
        pra = fma(r[c].x, y.x, fma(r[c].y, y.y, pra));
      }
      prr = wave_sum(prr);
      pra = wave_sum(pra);
      if (lane < 4) {
        const double val = lane == 0 ? prr : (lane == 1 ? pra : (lane == 2 ? s_r[2] : s_r[Rg + 1]));
        granule_put(X1 + ((int64_t)g * 4 + lane) * 2, epoch, val);
      }
    }
    for (int m = wv; m < 4; m += LZM_WAVES) {
      // the waves gather the records of workgroups 64 m + lane (wave 0 after publishing its own); every workgroup then adds the four wave sums in the
      // same order.  Cross terms r_last(gg) r_first(gg + 1): the next lane's record; the last lane of a wave takes the
      // first record of the next wave through LDS (s_b[8 + 4 m + 3] below).
      const int gg = lane + 64 * m;
      double rec[4] = {0.0, 0.0, 0.0, 0.0};
      if (gg < G) {
        for (;;) {
          bool ok = true;
#pragma unroll
          for (int f = 0; f < 4; ++f) ok &= granule_try_get(X1 + ((int64_t)gg * 4 + f) * 2, epoch, rec[f]);
          if (ok) break;
          __builtin_amdgcn_s_sleep(1);
          if (wall_clock64() - t0 > LZM_TIMEOUT_TICKS) {
            s_b[1] = 1.0;
            break;
          }
        }
      }
      const double nxt = __shfl_down(rec[2], 1);
      double cross = (lane < 63 && gg + 1 < G) ? rec[3] * nxt : 0.0;
      const double rr = wave_sum(rec[0]), ra = wave_sum(rec[1]);
      cross = wave_sum(cross);
      if (gg == g - 1) s_b[2] = rec[3];
      if (gg == g + 1) s_b[3] = rec[2];
      if (lane == 0) {
        s_b[8 + 4 * m + 0] = rr;
        s_b[8 + 4 * m + 1] = ra;
        s_b[8 + 4 * m + 2] = cross;
        s_b[8 + 4 * m + 3] = rec[2];              // r_first of workgroup 64 m
      }
      if (lane == 63) s_b[24 + m] = rec[3];       // r_last of workgroup 64 m + 63
    }
    __syncthreads();                                                             // B1
    LZM_TICK(0)
    if (s_b[1] != 0.0) {
      if (tid == 0) a.fail[0] = 1.0;
      return;
    }
    // the four wave sums in a fixed order; wrap-around cross terms between the waves' ranges
    const double rr_all = ((s_b[8] + s_b[12]) + s_b[16]) + s_b[20];
    double cross_all = ((s_b[10] + s_b[14]) + s_b[18]) + s_b[22];
#pragma unroll
    for (int m = 0; m < 3; ++m)
      if (64 * (m + 1) < G) cross_all = fma(s_b[24 + m], s_b[8 + 4 * (m + 1) + 3], cross_all);
    const double rAr_all = __dadd_rn(((s_b[9] + s_b[13]) + s_b[17]) + s_b[21], __dmul_rn(2.0 * coef, cross_all));
    const double hl_own = (g == 0) ? 0.0 : s_b[2], hr_own = (g == G - 1) ? 0.0 : s_b[3];   // Dirichlet ends (schrodinger1D.py:20-21)
    const double beta = sqrt(rr_all);
    if (s >= 1) {
      if (g == 0 && tid == 0) a.betas[s - 1] = beta;
      scale = fmax(scale, fabs(beta));
      if (!(beta > DSEA_BREAK_TOL * scale)) {            // also catches NaN; the same decision everywhere
        if (g == 0 && tid == 0) {
          a.brk[0] = (double)s;
          a.brk[1] = scale;
        }
        return;
      }
    }
    const double alpha = rAr_all / rr_all;
    if (g == 0 && tid == 0) a.alphas[s] = alpha;
    scale = fmax(scale, fabs(alpha));
    // ---------------------------------------------------------------- q_s = r / beta, u = A r / beta, three-term update
    double2 q[LZM_NCH], rn[LZM_NCH];
    double* s_qs = s_q + (int64_t)(s & 1) * R;           // q_s goes here (wave 3), q_{s-1} is in the other half
    {
      const double hl = hl_own, hr = hr_own;
      const double bprev = s >= 1 ? beta : 0.0;
      const double* s_qp = s_q + (int64_t)((s & 1) ^ 1) * R;
#pragma unroll
      for (int c = 0; c < LZM_NCH; ++c) {
        const int lr = 128 * c + 2 * lane;
        double2 y = make_double2(0.0, 0.0);
        if (lr < R) y = *reinterpret_cast<const double2*>(s_y + lr);
        // the two edge rows of the slab see their neighbours' rows (zero halos in s_y): recomputed in full
        if (lr == 0 && vx[c]) y.x = lzm_stencil_row(coef, s_V[0], r[c].x, vy[c] ? r[c].y : hr, hl);
        if (lr == Rg - 1) y.x = lzm_stencil_row(coef, s_V[lr], r[c].x, hr, lr > 0 ? s_r[lr + 1] : hl);
        if (lr + 1 == Rg - 1) y.y = lzm_stencil_row(coef, s_V[lr + 1], r[c].y, hr, r[c].x);
        q[c].x = r[c].x / beta;
        q[c].y = r[c].y / beta;
        const double ux = y.x / beta, uy = y.y / beta;
        double2 qp = make_double2(0.0, 0.0);
        if (lr < R) qp = *reinterpret_cast<const double2*>(s_qp + lr);
        rn[c].x = __dsub_rn(__dsub_rn(ux, __dmul_rn(alpha, q[c].x)), __dmul_rn(bprev, qp.x));
        rn[c].y = __dsub_rn(__dsub_rn(uy, __dmul_rn(alpha, q[c].y)), __dmul_rn(bprev, qp.y));
      }
    }
    // store q_s: basis row (wave 0), bf16 shadow row (wave 1), on-chip copy
    if (wv == 0) {
#pragma unroll
      for (int c = 0; c < LZM_NCH; ++c) {
        const int lr = 128 * c + 2 * lane;
        if (vy[c]) *reinterpret_cast<double2*>(a.Q + (int64_t)s * a.ldq + base + lr) = q[c];
        else if (vx[c]) a.Q[(int64_t)s * a.ldq + base + lr] = q[c].x;
      }
    } else if (wv == 1 && a.Qs) {
#pragma unroll
      for (int c = 0; c < LZM_NCH; ++c) {
        const int lr = 128 * c + 2 * lane;
        if (vx[c]) st_bf16x2(a.Qs + (int64_t)s * a.lds + base, lr, Rg, q[c]);
      }
    } else if (wv == 2 && s < NL) {
#pragma unroll
      for (int c = 0; c < LZM_NCH; ++c) {
        const int lr = 128 * c + 2 * lane;
        if (lr < R) *reinterpret_cast<double2*>(s_cache + (int64_t)s * R + lr) = q[c];
      }
    } else if (wv == 3) {
      // (the other half, q_{s-1}, was last read before B2 of the previous step; this half, q_{s-2}, before B4 of step s-2)
#pragma unroll
      for (int c = 0; c < LZM_NCH; ++c) {
        const int lr = 128 * c + 2 * lane;
        if (lr < R) *reinterpret_cast<double2*>(s_qs + lr) = q[c];
      }
    }
    if (s >= NL && s < M && ((s - NL) & (LZM_WAVES - 1)) == wv) {
      const int slot = (s - NL) / LZM_WAVES;
#pragma unroll
      for (int sl = 0; sl < LZM_NR; ++sl)
        if (sl == slot) {
#pragma unroll
          for (int c = 0; c < LZM_NCH; ++c) qc[sl][c] = q[c];
        }
    }
    if (s == a.k - 1) break;
    LZM_TICK(1)
    // ---------------------------------------------------------------- partial c_j = q_j . r'  (j <= s), ||r'||^2
    {
      // (a) this wave's register slots (zeros where nothing is cached yet: harmless)
#pragma unroll
      for (int s4 = 0; s4 < LZM_NR; s4 += 4) {
        double acc[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          acc[v] = 0.0;
          if (s4 + v < LZM_NR) {          // (compile-time: NR need not be a multiple of four)
#pragma unroll
            for (int c = 0; c < LZM_NCH; ++c) {
              const double2 qq = qc[s4 + v < LZM_NR ? s4 + v : 0][c];
              acc[v] = fma(qq.x, rn[c].x, fma(qq.y, rn[c].y, acc[v]));
            }
          }
        }
        const double bsum = wave_sum4_rows(acc[0], acc[1], acc[2], acc[3]);
        const int slot = s4 + (lane >> 4);
        const int j = NL + LZM_WAVES * slot + wv;
        if ((lane & 15) == 15 && slot < LZM_NR && j < s) s_cpart[j] = bsum;
      }
      // (b) LDS-cached vectors j < min(NL, s), (c) streamed vectors M <= j < s: chunks of four, round robin over the waves
      const int nlds = s < NL ? s : NL;
      for (int cc = wv; 4 * cc < nlds; cc += LZM_WAVES) {
        const int j0 = 4 * cc;
        double acc[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          acc[v] = 0.0;
          if (j0 + v < nlds) {
#pragma unroll
            for (int c = 0; c < LZM_NCH; ++c) {
              const int lr = 128 * c + 2 * lane;
              if (lr < R) {
                const double2 qq = *reinterpret_cast<const double2*>(s_cache + (int64_t)(j0 + v) * R + lr);
                acc[v] = fma(qq.x, rn[c].x, fma(qq.y, rn[c].y, acc[v]));
              }
            }
          }
        }
        const double bsum = wave_sum4_rows(acc[0], acc[1], acc[2], acc[3]);
        const int j = j0 + (lane >> 4);
        if ((lane & 15) == 15 && j < nlds) s_cpart[j] = bsum;
      }
      for (int cc = wv; M + 4 * cc < s; cc += LZM_WAVES) {
        const int j0 = M + 4 * cc;
        double acc[4];
        {                                           // four vectors (16 loads of 16 bytes per lane) in flight at a time
          double2 qq[4][LZM_NCH];
#pragma unroll
          for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int c = 0; c < LZM_NCH; ++c) {
              const int lr = 128 * c + 2 * lane;
              qq[v][c] = make_double2(0.0, 0.0);
              if (j0 + v < s && vx[c]) qq[v][c] = ld2<true>(a.Q + (int64_t)(j0 + v) * a.ldq + base, lr, Rg);
            }
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            acc[v] = 0.0;
#pragma unroll
            for (int c = 0; c < LZM_NCH; ++c) acc[v] = fma(qq[v][c].x, rn[c].x, fma(qq[v][c].y, rn[c].y, acc[v]));
          }
        }
        const double bsum = wave_sum4_rows(acc[0], acc[1], acc[2], acc[3]);
        const int j = j0 + (lane >> 4);
        if ((lane & 15) == 15 && j < s) s_cpart[j] = bsum;
      }
      // (d) j = s: the vector just formed; (e) ||r'||^2 (pseudo-vector s + 1: the premise of the shadow pass)
      if (wv == (s & (LZM_WAVES - 1))) {
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < LZM_NCH; ++c) acc = fma(q[c].x, rn[c].x, fma(q[c].y, rn[c].y, acc));
        acc = wave_sum(acc);
        if (lane == 0) s_cpart[s] = acc;
      }
      if (wv == ((s + 1) & (LZM_WAVES - 1))) {
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < LZM_NCH; ++c) acc = fma(rn[c].x, rn[c].x, fma(rn[c].y, rn[c].y, acc));
        acc = wave_sum(acc);
        if (lane == 0) s_cpart[s + 1] = acc;
      }
    }
    __syncthreads();                                                             // B2
    LZM_TICK(2)
    // ---------------------------------------------------------------- X2a: partials to their owners, X2b: reduced values to all
    const int nvec = s + 2;
    for (int t = tid; t < nvec; t += LZM_THREADS) granule_put(X2a + ((int64_t)t * G + g) * 2, epoch, s_cpart[t]);
    for (int j = g + wv * G; j < nvec; j += LZM_WAVES * G) {          // the coefficients this workgroup owns
      double v[4][1];
      if (!lzm_poll_records<1>(X2a + (int64_t)j * G * 2, G, lane, epoch, v, t0)) s_b[1] = 1.0;
      double tot = ((v[0][0] + v[1][0]) + v[2][0]) + v[3][0];
      tot = wave_sum(tot);
      if (lane == 0) granule_put(X2b + (int64_t)j * 2, epoch, tot);
    }
    LZM_TICK(3)
    for (int t = tid; t < nvec; t += LZM_THREADS) {
      double v = 0.0;
      while (!granule_try_get(X2b + (int64_t)t * 2, epoch, v)) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > LZM_TIMEOUT_TICKS) {
          s_b[1] = 1.0;
          break;
        }
      }
      s_c[t] = v;
    }
    if (tid == 0) s_b[5] = 0.0;
    __syncthreads();                                                             // B3
    LZM_TICK(4)
    if (s_b[1] != 0.0) {
      if (tid == 0) a.fail[0] = 1.0;
      return;
    }
    // premise of the shadow pass: max_j c_j^2 <= tau^2 ||r'||^2 (identical decision everywhere: same scalars)
    const bool stream_any = s > M;
    bool use_shadow = a.Qs != nullptr && stream_any;
    if (use_shadow) {
      const double lim = a.tau * a.tau * s_c[s + 1];
      for (int t = tid; t <= s; t += LZM_THREADS)
        if (!(s_c[t] * s_c[t] <= lim)) s_b[5] = 1.0;
      __syncthreads();
      use_shadow = s_b[5] == 0.0;
    }
    if (stream_any && !use_shadow) ++fb_steps; else ++lp_steps;
    LZM_TICK(5)
    // ---------------------------------------------------------------- r = r' - sum_{j<=s} c_j q_j
    {
      double2 w[LZM_NCH];
#pragma unroll
      for (int c = 0; c < LZM_NCH; ++c) w[c] = make_double2(0.0, 0.0);
#pragma unroll
      for (int sl = 0; sl < LZM_NR; ++sl) {
        const int j = NL + LZM_WAVES * sl + wv;
        const double cj = (j < s) ? s_c[j] : 0.0;
#pragma unroll
        for (int c = 0; c < LZM_NCH; ++c) {
          const double2 qq = qc[sl][c];
          w[c].x = fma(cj, qq.x, w[c].x);
          w[c].y = fma(cj, qq.y, w[c].y);
        }
      }
      const int nlds = s < NL ? s : NL;
      for (int j = wv; j < nlds; j += LZM_WAVES) {
        const double cj = s_c[j];
#pragma unroll
        for (int c = 0; c < LZM_NCH; ++c) {
          const int lr = 128 * c + 2 * lane;
          if (lr < R) {
            const double2 qq = *reinterpret_cast<const double2*>(s_cache + (int64_t)j * R + lr);
            w[c].x = fma(cj, qq.x, w[c].x);
            w[c].y = fma(cj, qq.y, w[c].y);
          }
        }
      }
      if (wv == (s & (LZM_WAVES - 1))) {
        const double cj = s_c[s];
#pragma unroll
        for (int c = 0; c < LZM_NCH; ++c) {
          const int lr = 128 * c + 2 * lane;
          if (lr < R) {
            const double2 qq = *reinterpret_cast<const double2*>(s_qs + lr);
            w[c].x = fma(cj, qq.x, w[c].x);
            w[c].y = fma(cj, qq.y, w[c].y);
          }
        }
      }
      double* mine = s_part + (int64_t)wv * R;
      if (stream_any && use_shadow) {
        // streamed vectors from the bf16 shadow: lane l covers the 8 consecutive rows 8 l .. 8 l + 7 of the slab (one
        // 16-byte load per vector); 4 vectors per trip, round robin over the waves
        double ws8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) ws8[e] = 0.0;
        const bool act = 8 * lane < Rg;          // (R and the slab bases are multiples of 8; a ragged tail reads zeros... see launcher)
        for (int cc = wv; M + 8 * cc < s; cc += LZM_WAVES) {
          const int j0 = M + 8 * cc;
          uint4 pk[8];
          double cj[8];
#pragma unroll
          for (int v = 0; v < 8; ++v) {
            pk[v] = make_uint4(0u, 0u, 0u, 0u);
            cj[v] = 0.0;
            if (j0 + v < s) {
              cj[v] = s_c[j0 + v];
              if (act) pk[v] = ld_u4_stream(a.Qs + (int64_t)(j0 + v) * a.lds + base + 8 * lane);
            }
          }
#pragma unroll
          for (int v = 0; v < 8; ++v) {
            ws8[0] = fma(cj[v], bf16lo_to_f64(pk[v].x), ws8[0]);
            ws8[1] = fma(cj[v], bf16hi_to_f64(pk[v].x), ws8[1]);
            ws8[2] = fma(cj[v], bf16lo_to_f64(pk[v].y), ws8[2]);
            ws8[3] = fma(cj[v], bf16hi_to_f64(pk[v].y), ws8[3]);
            ws8[4] = fma(cj[v], bf16lo_to_f64(pk[v].z), ws8[4]);
            ws8[5] = fma(cj[v], bf16hi_to_f64(pk[v].z), ws8[5]);
            ws8[6] = fma(cj[v], bf16lo_to_f64(pk[v].w), ws8[6]);
            ws8[7] = fma(cj[v], bf16hi_to_f64(pk[v].w), ws8[7]);
          }
        }
        // change of lane layout through this wave's own LDS row: write the 8-row form, read back the pair form
        if (8 * lane < R) {
#pragma unroll
          for (int e = 0; e < 8; e += 2) *reinterpret_cast<double2*>(mine + 8 * lane + e) = make_double2(ws8[e], ws8[e + 1]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < LZM_NCH; ++c) {
          const int lr = 128 * c + 2 * lane;
          if (lr < R) {
            const double2 t = *reinterpret_cast<const double2*>(mine + lr);
            w[c].x += t.x;
            w[c].y += t.y;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
      } else if (stream_any) {
        for (int cc = wv; M + 2 * cc < s; cc += LZM_WAVES) {
          const int j0 = M + 2 * cc;
          double2 qq[2][LZM_NCH];
          double cj[2];
#pragma unroll
          for (int v = 0; v < 2; ++v) {
            cj[v] = (j0 + v < s) ? s_c[j0 + v] : 0.0;
#pragma unroll
            for (int c = 0; c < LZM_NCH; ++c) {
              const int lr = 128 * c + 2 * lane;
              qq[v][c] = make_double2(0.0, 0.0);
              if (j0 + v < s && vx[c]) qq[v][c] = ld2<true>(a.Q + (int64_t)(j0 + v) * a.ldq + base, lr, Rg);
            }
          }
#pragma unroll
          for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int c = 0; c < LZM_NCH; ++c) {
              w[c].x = fma(cj[v], qq[v][c].x, w[c].x);
              w[c].y = fma(cj[v], qq[v][c].y, w[c].y);
            }
        }
      }
#pragma unroll
      for (int c = 0; c < LZM_NCH; ++c) {
        const int lr = 128 * c + 2 * lane;
        if (lr < R) *reinterpret_cast<double2*>(mine + lr) = w[c];
      }
      LZM_TICK(6)
      __syncthreads();                                                           // B4
#pragma unroll
      for (int c = 0; c < LZM_NCH; ++c) {
        const int lr = 128 * c + 2 * lane;
        double2 tot = make_double2(0.0, 0.0);
        if (lr < R) {
          tot = *reinterpret_cast<const double2*>(s_part + lr);
#pragma unroll
          for (int k2 = 1; k2 < LZM_WAVES; ++k2) {
            const double2 t = *reinterpret_cast<const double2*>(s_part + (int64_t)k2 * R + lr);
            tot.x += t.x;
            tot.y += t.y;
          }
        }
        r[c].x = vx[c] ? rn[c].x - tot.x : 0.0;
        r[c].y = vy[c] ? rn[c].y - tot.y : 0.0;
      }
    }
    LZM_TICK(7)
    // no barrier here: s_part / s_c are next written behind B1..B3 of the next step, s_r / s_y by wave 0 after it has
    // passed B4 (every wave is then beyond its reads of them), s_b by wave 1 likewise
  }
#ifdef DSEA_LZM_TIMING
  if (g == 7 % G && tid == 0)
    for (int q = 0; q < 8; ++q) a.alphas[q] = (double)tacc[q] * 0.01 / (double)a.k;   // us per step (overwrites alphas!)
#endif
  if (g == 0 && tid == 0) {
    a.brk[1] = scale;
    if (a.lp_count) {
      a.lp_count[0] = (double)lp_steps;
      a.lp_count[1] = (double)fb_steps;
    }
  }
}

namespace {
// slab rows: ceil(n / G) rounded up to a multiple of 8 (16-byte aligned fp64 pairs and 16-byte bf16 octets)
inline int lzm_rows(int64_t n, int cus, int* G_out) {
  int G = cus < LZM_MAX_G ? cus : LZM_MAX_G;
  int64_t R = (n + G - 1) / G;
  R = (R + 7) / 8 * 8;
  if (R < 64) R = 64;
  *G_out = (int)((n + R - 1) / R);
  return (int)R;
}
inline size_t lzm_lds_doubles(int R, int NL) {
  return (size_t)2 * LZM_K2 + 32 + (size_t)(R + 4) + (size_t)R * 4 + (size_t)LZM_WAVES * R + (size_t)NL * R;
}
inline int lzm_cus() {
  static thread_local int cu_dev = -1, cu_count = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (dev != cu_dev) {
    if (hipDeviceGetAttribute(&cu_count, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    cu_dev = dev;
  }
  return cu_count;
}
}  // namespace

// Is the mid-size single-launch form applicable?  (halo-free 3-point stencil, 8192 < n <= G x 512 rows, 2 <= k <= 512)
bool lanczos_persist_mid_applicable(const OpDesc& op, int64_t n, int k) {
  if (op.kind != OP_STENCIL3 || op.st3.halo_lo || op.st3.halo_hi) return false;
  if (k < 2 || k > LZM_MAX_K || n <= 8192) return false;
  const int cus = lzm_cus();
  if (cus < 1) return false;
  int G = 0;
  const int R = lzm_rows(n, cus, &G);
  return R <= 128 * LZM_NCH && G <= cus && G <= LZM_MAX_G;
}

size_t lanczos_persist_mid_comm_bytes(int64_t n, int k) {
  int G = 0;
  const int cus = lzm_cus();
  lzm_rows(n, cus < 1 ? LZM_MAX_G : cus, &G);
  const int64_t kslots = (int64_t)k + 2;
  return (size_t)(2 * ((int64_t)G * 4 + kslots * G + kslots)) * sizeof(unsigned long long);   // X1 | X2a | X2b
}

// returns 0 if launched, -1 if not applicable (caller runs the multi-launch form), -2 on a HIP error
int launch_lanczos_persist_mid(const OpDesc& op, int k, const double* q0, double* Q, int64_t ldq, uint16_t* Qs, int64_t lds,
                               double tau, double* alphas, double* betas, double* brk, double* fail, double* lp_count,
                               void* comm, hipStream_t st, int lose_peer) {
  const int64_t n = op.n;
  if (!lanczos_persist_mid_applicable(op, n, k)) return -1;
  const int cus = lzm_cus();
  int G = 0;
  const int R = lzm_rows(n, cus, &G);
  // the 8-row octets of the shadow pass read up to the end of the slab's last octet: inside the row as long as the row
  // stride covers the padded slab
  if (Qs && lds < (int64_t)(G - 1) * R + ((n - (int64_t)(G - 1) * R + 7) / 8 * 8)) Qs = nullptr;
  static thread_local int attr_done_dev = -1;
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -2;
    if (dev != attr_done_dev) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_lanczos_persist_mid), hipFuncAttributeMaxDynamicSharedMemorySize,
                              LZM_LDS_BYTES) != hipSuccess)
        return -2;
      attr_done_dev = dev;
    }
  }
  // LDS-cached vectors: what is left of the CU's 160 KiB behind the fixed arrays
  int NL = (int)((LZM_LDS_BYTES / sizeof(double) - lzm_lds_doubles(R, 0)) / (size_t)R);
  if (NL > k) NL = k;
  NL = NL / 4 * 4;
  const size_t lds_bytes = lzm_lds_doubles(R, NL) * sizeof(double);
  int occ = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_lanczos_persist_mid, LZM_THREADS, lds_bytes) != hipSuccess || occ < 1)
    return -1;
  if (hipMemsetAsync(comm, 0, lanczos_persist_mid_comm_bytes(n, k), st) != hipSuccess) return -2;
  LzmArgs a;
  a.st = op.st3;
  a.q0 = q0;
  a.Q = Q;
  a.ldq = ldq;
  a.n = n;
  a.k = k;
  a.Qs = Qs;
  a.lds = lds;
  a.tau = tau;
  a.alphas = alphas;
  a.betas = betas;
  a.brk = brk;
  a.fail = fail;
  a.lp_count = lp_count;
  a.comm = static_cast<unsigned long long*>(comm);
  a.G = G;
  a.R = R;
  a.kslots = k + 2;
  a.NL = NL;
  a.lose_peer = lose_peer;
  hipLaunchKernelGGL(k_lanczos_persist_mid, dim3(G), dim3(LZM_THREADS), lds_bytes, st, a);
  return 0;
}

}  // namespace dsea
