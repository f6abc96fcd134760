// dsea_lanczos_persist.hip -- the k-step Lanczos loop of reference Lanczos.py:49-77 as ONE launch, for the problem
// sizes the reference's own README, tests and examples use (n <= 8192 rows: examples/TFIM/E0.py N = 10..13,
// tests/test_Lanczos.py n = 1000, examples/schrodinger1D.py N = 300).
//
// Why: there the streaming form is bound by the dependency latency of its five launches per step (dots, coefficient
// finalisation, correction, normalise + mat-vec: 16-19 us per step at L = 8 ... 12 on MI355X, every kernel a few us of
// dependent memory round trips behind a 1.5-1.9 us launch boundary).  A grid-wide exchange inside a launch costs about
// what a boundary costs (~2.4 us, measured in k_cg_persist_stencil), so a persistent kernel wins only what the kernels'
// own prologues and the second reduction stage cost.  Per step, three exchanges -- the reference's own data flow:
//
//     [E3: ||r||^2 partials + the rows of r the neighbours' mat-vec needs]
//     q_s = r / ||r||  ;  u = A q_s                       (slab-local: every workgroup owns 128 rows)  Lanczos.py:69-71
//     [EA: alpha_s = q_s . u]                                                                           Lanczos.py:72
//     r = u - alpha_s q_s - beta_{s-1} q_{s-1}  ;  partial c_j = q_j . r  for j <= s                   Lanczos.py:61,66
//     [E2: all-reduce of the s + 1 coefficients]
//     r -= sum_{j<=s} c_j q_j                                                                           Lanczos.py:66
//
// A TWO-exchange formulation was built first and measured (classical Gram-Schmidt of u against all basis vectors in one
// projection, r = u - Q Q^T u, alpha_s = (Q^T u)_s: 9.2 / 11.2 / 16.2 us per step at L = 8 / 10 / 12 against 16.5 / 18.4 /
// 20.4) and REJECTED: it is numerically weaker exactly where Lanczos needs strength.  The reference's order removes the
// two O(1) components with the known alpha, beta first and lets the Gram-Schmidt pass measure only rounding-level
// residue of a SMALL vector (coefficient error ~ eps ||r||); the one-projection form measures everything against the
// large u (error ~ eps ||u||), loses orthogonality like eps ||u|| / beta and, once Ritz values converge and beta drops,
// amplifies it step after step: TFIM L = 8 with k = 200 (beyond the Krylov dimension of the start vector) returned
// E0 = -39.8 against -10.2517 while the reference's order sails through (beta down to 9e-6, residual 2e-15).
// Arithmetic here follows the multi-launch kernels expression by expression; what differs is the ORDER in which partial
// sums are combined (alpha per 128-row slab instead of per 2048-row mat-vec tile), so T agrees with the multi-launch
// form to rounding, not bit for bit.  tests/test_gpu_persistent.py holds it to the oracle and to that form.
//
// Mechanics: G = ceil(n / 128) <= 64 workgroups of 1024 threads, all co-resident (one per CU); a workgroup keeps no
// cross-workgroup state but what it reads from two granule buffers (8-byte data + 32-bit epoch tag per 8-byte word,
// relaxed agent-scope stores / polls, no fences: cdna_hip_programming.md Guideline 16 form R2, as in the persistent
// CG).  Each workgroup re-reads only ITS OWN rows of the basis (plain global memory, written by itself: no
// cross-workgroup coherence involved), 16 waves split the basis vectors.  Every workgroup sums the same partials in the
// same order, so all of them hold bit-identical coefficients / norms and take the same breakdown decision.  Spins are
// bounded by a wall-clock timeout (a lost peer must not hang the GPU): the launch then sets a fail flag, the host
// repeats the run with the streaming kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dsea_internal.h"
#include "dsea_device.h"

namespace dsea {

namespace {
typedef gran_u64 lzp_gu64;
#define LZP_TIMEOUT_TICKS 30000000ll   /* 0.3 s of the 100 MHz wall clock: a healthy step takes ~10 us, a lost peer must not hold
                                          up to 64 CUs for seconds before the host falls back to the multi-launch kernels */
#define LZP_ROWS 128
#define LZP_MAX_K 512
#define LZP_MAX_G 64
#define LZP_CACHE 112   // basis vectors whose own rows stay in LDS (112 KB of the CU's 160; a multiple of the chunk of 4)

// spin until the granule carries `epoch`; false on timeout
__device__ __forceinline__ bool lzp_wait(lzp_gu64* g, unsigned epoch, double& v, long long t0) {
  while (!granule_try_get(g, epoch, v)) {
    __builtin_amdgcn_s_sleep(1);
    if (wall_clock64() - t0 > LZP_TIMEOUT_TICKS) return false;
  }
  return true;
}

__device__ __forceinline__ double lzp_stencil_row(double coef, double Vi, double xi, double up, double dn) {
  const double lap = __dadd_rn(__dadd_rn(__dmul_rn(-2.0, xi), up), dn);
  return __dadd_rn(__dmul_rn(coef, lap), __dmul_rn(Vi, xi));
}
__device__ __forceinline__ double lzp_tfim_diag(const TfimParams& p, int64_t i, uint64_t maskL) {
  const uint64_t gi = (uint64_t)(p.row_offset + i);
  const uint64_t rot = ((gi << 1) | (gi >> (p.L - 1))) & maskL;
  const int pop = __popcll(gi ^ rot);
  return p.diag_scale * (double)(-(p.L - 2 * pop));
}
}  // namespace

struct LzpArgs {
  int opk;  // OP_TFIM / OP_STENCIL3
  TfimParams tf;
  Stencil3Params st;
  const double* q0;
  double* Q;
  int64_t ldq, n;
  int k;
  double* alphas;
  double* betas;
  double* brk;   // [0] breakdown step, [1] running scale
  double* fail;  // set to 1 when a peer did not arrive in time
  unsigned long long* comm;
  int G, kslots;
  int lose_peer;   // test hook: the last workgroup exits at once (its peers must time out, not hang)
};

__global__ __launch_bounds__(1024) void k_lanczos_persist(LzpArgs a) {
  __shared__ double s_c[LZP_MAX_K];          // reduced coefficients (identical in every workgroup)
  __shared__ double s_cpart[LZP_MAX_K];      // this workgroup's partial dots
  __shared__ double s_red[1024];             // gather: partial sums over workgroup ranges, [part * J + j]
  __shared__ double2 s_part[16][64];         // correction pass: per-wave partial sums
  __shared__ double2 s_cache[LZP_CACHE][64]; // own rows of q_0 .. q_{LZP_CACHE-1}: the two passes over the basis read these from LDS
  __shared__ double s_q[LZP_ROWS];           // own rows of q_s
  __shared__ double s_qp[LZP_ROWS];          // own rows of q_{s-1}
  __shared__ double s_u[LZP_ROWS];           // own rows of u = A q_s
  __shared__ double s_nb[6][LZP_ROWS];       // TFIM: the rows of the (un-normalised) r of the up-to-6 partner workgroups
  __shared__ double s_alpha;                 // alpha_s (s_b[0] still holds ||r||^2 when it is written)
  __shared__ double s_b[4];                  // [0] ||r||^2  [1] fail  [2] left edge  [3] right edge (stencil)

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int g = blockIdx.x, G = a.G;
  if (a.lose_peer && G > 1 && g == G - 1) return;
  const int64_t n = a.n, base = (int64_t)g * LZP_ROWS, row = base + 2 * lane;
  const bool tfim = a.opk == OP_TFIM;
  const int Lbits = tfim ? a.tf.L : 0;
  const int nlocal = Lbits < 7 ? Lbits : 7;     // bit flips inside the 128-row slab
  const int nfar = tfim ? Lbits - nlocal : 0;   // partner workgroups g ^ (1 << b)
  const uint64_t maskL = (Lbits >= 64) ? ~0ull : ((1ull << Lbits) - 1ull);
  lzp_gu64* E2 = (lzp_gu64*)a.comm;
  lzp_gu64* E3 = E2 + (int64_t)2 * G * a.kslots;
  lzp_gu64* EA = E3 + (int64_t)2 * G * (1 + LZP_ROWS);
  if (tid == 0) s_b[1] = 0.0;
  if (tid < LZP_ROWS) s_q[tid] = 0.0;
  double scale = 0.0;                           // running max |alpha|, |beta| (same in every thread of every workgroup)
  double2 rv = make_double2(0.0, 0.0);          // wave 0: this lane's two rows of r
  if (wv == 0) rv = ld2<true>(a.q0, row, n);
  __syncthreads();

#ifdef DSEA_LZP_TIMING
  long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = wall_clock64();
#define LZP_TICK(k) { const long long tn = wall_clock64(); tacc[k] += tn - tprev; tprev = tn; }
#else
#define LZP_TICK(k)
#endif
  for (int s = 0; s < a.k; ++s) {
    const unsigned epoch = (unsigned)(s + 1);
    const long long t0 = wall_clock64();
    // ---- E3: publish ||r||^2 partial and the rows of r; gather all partials (+ the partner rows / edge rows)
    if (wv == 0) {
      const double acc = wave_sum(fma(rv.x, rv.x, rv.y * rv.y));
      lzp_gu64* mine = E3 + (int64_t)g * (1 + LZP_ROWS) * 2;
      granule_put(mine + (1 + 2 * lane) * 2, epoch, rv.x);
      granule_put(mine + (2 + 2 * lane) * 2, epoch, rv.y);
      if (lane == 0) granule_put(mine, epoch, acc);
    } else if (wv == 1) {
      // the G slab partials, one per lane, summed by the wave itself in its fixed order (identical in every workgroup)
      double v = 0.0;
      if (lane < G && !lzp_wait(E3 + (int64_t)lane * (1 + LZP_ROWS) * 2, epoch, v, t0)) s_b[1] = 1.0;
      v = wave_sum(v);
      if (lane == 0) s_b[0] = v;
    }
    if (tfim) {
      const int t = tid - 128;
      if (t >= 0 && t < nfar * LZP_ROWS) {
        const int b = t >> 7, rr = t & 127;
        double v = 0.0;
        if (!lzp_wait(E3 + ((int64_t)(g ^ (1 << b)) * (1 + LZP_ROWS) + 1 + rr) * 2, epoch, v, t0)) s_b[1] = 1.0;
        s_nb[b][rr] = v;
      }
    } else if (tid == 128 || tid == 192) {
      const bool left = tid == 128;
      const int peer = left ? g - 1 : g + 1;
      double v = 0.0;
      if (peer >= 0 && peer < G)
        if (!lzp_wait(E3 + ((int64_t)peer * (1 + LZP_ROWS) + 1 + (left ? LZP_ROWS - 1 : 0)) * 2, epoch, v, t0)) s_b[1] = 1.0;
      s_b[left ? 2 : 3] = v;
    }
    __syncthreads();
    if (s_b[1] != 0.0) {
      if (tid == 0) a.fail[0] = 1.0;
      return;
    }
    const double beta = sqrt(s_b[0]);
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
    LZP_TICK(0)
    // ---- q_s = r / beta (Lanczos.py:53,70), stored into the basis; u = A q_s on this slab
    if (wv == 0) {
      double2 q;
      q.x = rv.x / beta;
      q.y = rv.y / beta;
      st2<true>(a.Q + (int64_t)s * a.ldq, row, n, q);
      if (s < LZP_CACHE) s_cache[s][lane] = q;
      s_qp[2 * lane] = s_q[2 * lane];          // q_{s-1} (wave 0 is the only writer of both arrays)
      s_qp[2 * lane + 1] = s_q[2 * lane + 1];
      s_q[2 * lane] = q.x;
      s_q[2 * lane + 1] = q.y;
      // wave 0 alone writes and (until the barrier after EA) reads s_q: the wave's own LDS operations complete in order,
      // the fence only stops the compiler from moving the reads above the writes
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
      double2 uu = make_double2(0.0, 0.0);
      const double x0 = s_q[2 * lane], x1 = s_q[2 * lane + 1];
      if (tfim) {
        const double gpar = a.tf.g_dev ? a.tf.g_dev[0] : a.tf.g_const;
        double s0 = 0.0, s1 = 0.0;
        for (int b = 0; b < nlocal; ++b) {
          s0 += s_q[(2 * lane) ^ (1 << b)];
          s1 += s_q[(2 * lane + 1) ^ (1 << b)];
        }
        double f0 = 0.0, f1 = 0.0;
        for (int b = 0; b < nfar; ++b) {
          f0 += s_nb[b][2 * lane];
          f1 += s_nb[b][2 * lane + 1];
        }
        if (nfar > 0) {   // the partners' rows are rows of the un-normalised r: scaled once (linearity)
          s0 += f0 / beta;
          s1 += f1 / beta;
        }
        if (row < n) uu.x = __dsub_rn(__dmul_rn(x0, lzp_tfim_diag(a.tf, row, maskL)), __dmul_rn(gpar, s0));
        if (row + 1 < n) uu.y = __dsub_rn(__dmul_rn(x1, lzp_tfim_diag(a.tf, row + 1, maskL)), __dmul_rn(gpar, s1));
      } else {
        const double2 Vv = ld2<true>(a.st.V, row, n);
        const double dn = lane > 0 ? s_q[2 * lane - 1] : s_b[2] / beta;                 // x[row - 1]
        const double up = lane < 63 ? s_q[2 * lane + 2] : s_b[3] / beta;                // x[row + 2]
        if (row < n) uu.x = lzp_stencil_row(a.st.coef, Vv.x, x0, (row + 1 < n) ? x1 : 0.0, dn);
        if (row + 1 < n) uu.y = lzp_stencil_row(a.st.coef, Vv.y, x1, (row + 2 < n) ? up : 0.0, x0);
      }
      // ---- EA: alpha_s = q_s . u  (Lanczos.py:72): slab partial published, everybody's gathered below
      const double pa = wave_sum(fma(x0, uu.x, x1 * uu.y));
      if (lane == 0) granule_put(EA + (int64_t)g * 2, epoch, pa);
      s_u[2 * lane] = uu.x;
      s_u[2 * lane + 1] = uu.y;
    } else if (wv == 1) {
      double v = 0.0;
      if (lane < G && !lzp_wait(EA + (int64_t)lane * 2, epoch, v, t0)) s_b[1] = 1.0;
      v = wave_sum(v);
      if (lane == 0) s_alpha = v;
    }
    __syncthreads();
    if (s_b[1] != 0.0) {
      if (tid == 0) a.fail[0] = 1.0;
      return;
    }
    const double alpha = s_alpha;
    if (g == 0 && tid == 0) a.alphas[s] = alpha;
    scale = fmax(scale, fabs(alpha));
    if (s == a.k - 1) break;
    LZP_TICK(1)
    // ---- three-term recurrence on the slab (Lanczos.py:61): r = u - alpha_s q_s - beta_{s-1} q_{s-1}, formed by EVERY
    // wave for its lanes' two rows (each wave covers all 128 rows): no hand-over through LDS, no barrier
    double2 uu;
    {
      const double b = s >= 1 ? beta : 0.0;
      uu.x = __dsub_rn(__dsub_rn(s_u[2 * lane], __dmul_rn(alpha, s_q[2 * lane])), __dmul_rn(b, s_qp[2 * lane]));
      uu.y = __dsub_rn(__dsub_rn(s_u[2 * lane + 1], __dmul_rn(alpha, s_q[2 * lane + 1])), __dmul_rn(b, s_qp[2 * lane + 1]));
    }
    // ---- partial c_j = q_j . r for j <= s (first half of Lanczos.py:66): chunks of four vectors, chunk cc -> wave cc mod 16
    const int nvec = s + 1;
    const int nchunks = (nvec + 3) / 4;
    for (int cc = wv; cc < nchunks; cc += 16) {
      const int j = 4 * cc;
      double acc[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        double2 q = make_double2(0.0, 0.0);
        if (j + v < nvec) q = (j < LZP_CACHE) ? s_cache[j + v][lane] : ld2<true>(a.Q + (int64_t)(j + v) * a.ldq, row, n);
        acc[v] = fma(q.x, uu.x, q.y * uu.y);
      }
      const double bsum = wave_sum4_rows(acc[0], acc[1], acc[2], acc[3]);
      const int jj = j + (lane >> 4);
      if ((lane & 15) == 15 && jj < nvec) s_cpart[jj] = bsum;
    }
    __syncthreads();
    LZP_TICK(2)
    // ---- E2: publish the partials, gather everybody's, sum over workgroups in a fixed order
    if (tid < nvec) granule_put(E2 + ((int64_t)g * a.kslots + tid) * 2, epoch, s_cpart[tid]);
    {
      // thread t -> coefficient j = t mod J, workgroup range `part` = t / J (J = power of two >= nvec): every thread polls at
      // most ceil(G / parts) granules, all in flight together; the ranges are then added in ascending order
      int J = 16;
      while (J < nvec) J <<= 1;
      const int parts = 1024 / J, Gp = (G + parts - 1) / parts;
      const int j = tid & (J - 1), part = tid / J;
      double acc = 0.0;
      if (j < nvec) {
        const int w0 = part * Gp, w1 = (w0 + Gp < G) ? w0 + Gp : G;
        const long long t1 = wall_clock64();
        for (int wb = w0; wb < w1; wb += 8) {
          double pv[8];
          bool ok;
          do {
            ok = true;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
              pv[m] = 0.0;
              if (wb + m < w1) ok &= granule_try_get(E2 + ((int64_t)(wb + m) * a.kslots + j) * 2, epoch, pv[m]);
            }
            if (!ok) {
              __builtin_amdgcn_s_sleep(1);
              if (wall_clock64() - t1 > LZP_TIMEOUT_TICKS) {
                s_b[1] = 1.0;
                break;
              }
            }
          } while (!ok);
#pragma unroll
          for (int m = 0; m < 8; ++m) acc += pv[m];      // absent ones are 0
        }
      }
      s_red[tid] = acc;
      __syncthreads();
      if (s_b[1] != 0.0) {
        if (tid == 0) a.fail[0] = 1.0;
        return;
      }
      if (tid < nvec) {
        double tot = s_red[tid];
        for (int pp = 1; pp < parts; ++pp) tot += s_red[pp * J + tid];
        s_c[tid] = tot;
      }
    }
    __syncthreads();
    LZP_TICK(3)
    // ---- r -= sum_{j<=s} c_j q_j   (second half of Lanczos.py:66), the waves' partial sums combined in wave order
    double2 w = make_double2(0.0, 0.0);
    for (int cc = wv; cc < nchunks; cc += 16) {
      const int j = 4 * cc;
      double2 q[4];
      double cj[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        q[v] = make_double2(0.0, 0.0);
        cj[v] = 0.0;
        if (j + v < nvec) {
          q[v] = (j < LZP_CACHE) ? s_cache[j + v][lane] : ld2<true>(a.Q + (int64_t)(j + v) * a.ldq, row, n);
          cj[v] = s_c[j + v];
        }
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        w.x = fma(cj[v], q[v].x, w.x);
        w.y = fma(cj[v], q[v].y, w.y);
      }
    }
    s_part[wv][lane] = w;
    __syncthreads();
    if (wv == 0) {
      double2 tot = s_part[0][lane];
#pragma unroll
      for (int k2 = 1; k2 < 16; ++k2) {
        tot.x += s_part[k2][lane].x;
        tot.y += s_part[k2][lane].y;
      }
      rv.x = uu.x - tot.x;
      rv.y = uu.y - tot.y;
    }
    // no barrier here: s_part / s_c / s_cpart / s_red are next written behind the next step's E3 and EA barriers
    LZP_TICK(4)
  }
#ifdef DSEA_LZP_TIMING
  if (g == 0 && tid == 0)
    for (int q = 0; q < 5; ++q) a.alphas[q] = (double)tacc[q] * 0.01 / (double)a.k;   // us per step (overwrites alphas!)
#endif
  if (g == 0 && tid == 0) a.brk[1] = scale;
}

// Is the single-launch form applicable?  (full-space TFIM or halo-free 3-point stencil, n <= 64 x 128 rows, k <= 512)
bool lanczos_persist_applicable(const OpDesc& op, int64_t n, int k) {
  if (k < 1 || k > LZP_MAX_K || n < 1 || n > (int64_t)LZP_MAX_G * LZP_ROWS) return false;
  if (op.kind == OP_TFIM)
    return op.tfim.L_local == op.tfim.L && op.tfim.row_offset == 0 && op.tfim.L >= 1 && op.tfim.L <= 13;
  if (op.kind == OP_STENCIL3) return !op.st3.halo_lo && !op.st3.halo_hi;
  return false;
}

size_t lanczos_persist_comm_bytes(int64_t n, int k) {
  const int64_t G = (n + LZP_ROWS - 1) / LZP_ROWS;
  return (size_t)(2 * G * ((int64_t)k + 1 + 1 + LZP_ROWS + 1)) * sizeof(unsigned long long);   // E2 | E3 | EA
}

// returns 0 if launched, -1 if not applicable (caller runs the streaming form), -2 on a HIP error
int launch_lanczos_persist(const OpDesc& op, int k, const double* q0, double* Q, int64_t ldq, double* alphas,
                           double* betas, double* brk, double* fail, void* comm, hipStream_t st, int lose_peer) {
  const int64_t n = op.n;
  if (!lanczos_persist_applicable(op, n, k)) return -1;
  const int G = (int)((n + LZP_ROWS - 1) / LZP_ROWS);
  {
    // all G workgroups must be resident together: 1024 threads and ~153 KB of static LDS each (s_cache 112 KB, s_part 16 KB,
    // s_red 8 KB, s_c / s_cpart 8 KB, s_nb 6 KB ...), i.e. one per CU and nothing else holding LDS there -- the occupancy
    // query below refuses the form otherwise (the host then takes the multi-launch kernels at once instead of timing out)
    static thread_local int cu_dev = -1, cu_count = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -2;
    if (dev != cu_dev) {
      if (hipDeviceGetAttribute(&cu_count, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -2;
      cu_dev = dev;
    }
    if (G > cu_count) return -1;
    static thread_local int occ_dev = -1, occ = 0;
    if (dev != occ_dev) {
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_lanczos_persist, 1024, 0) != hipSuccess) return -2;
      occ_dev = dev;
    }
    if (occ < 1) return -1;
  }
  if (hipMemsetAsync(comm, 0, lanczos_persist_comm_bytes(n, k), st) != hipSuccess) return -2;
  LzpArgs a;
  a.opk = (int)op.kind;
  a.tf = op.tfim;
  a.st = op.st3;
  a.q0 = q0;
  a.Q = Q;
  a.ldq = ldq;
  a.n = n;
  a.k = k;
  a.alphas = alphas;
  a.betas = betas;
  a.brk = brk;
  a.fail = fail;
  a.comm = static_cast<unsigned long long*>(comm);
  a.G = G;
  a.kslots = k + 1;
  a.lose_peer = lose_peer;
  hipLaunchKernelGGL(k_lanczos_persist, dim3(G), dim3(1024), 0, st, a);
  return 0;
}

}  // namespace dsea
