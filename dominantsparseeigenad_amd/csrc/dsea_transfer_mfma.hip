// dsea_transfer_mfma.hip -- the transfer-matrix mat-vec of reference examples/TFIM_vumps/general.py:59-66,
//     y = sum_s B_s X B_s^T        (X = x as a D x D row-major matrix, B = A or its slice-wise transpose),
// on the fp64 matrix cores (v_mfma_f64_16x16x4_f64) -- the one GEMM-shaped operand of the path (BASELINE
// configs[3]: D = 512, d = 2).
//
// Until round 3 this was four launches: transpose of x, two strided-batched rocBLAS DGEMMs, slice sum (32.7 us at D = 512).
// Here two hand-written kernels and no vendor library:
//   K1   T = [B_0; B_1; ...] X          one (dD x D)(D x D) product: the d slices are d D more rows of ONE row-major matrix
//   K2   y = [T_0 T_1 ...] [B_0 B_1 ...]^T   one product with inner dimension dD: the sum over s IS the inner sum, so the
//        result is written once -- no slice sum, and because both operands are read as they lie in memory (K2 is an
//        "A B^T" product: rows of T_s and rows of B_s are both contiguous along the inner index) no transpose either.
// Tiling for 256 CUs, not for a big-GEMM library shape: the output has only D^2 = 2^18 elements, so a workgroup takes a
// 64 x 32 (K1: 16 x 16 = 256 workgroups) / 32 x 32 (K2: 256 workgroups) tile -- one workgroup per CU, one wave per SIMD.
// k_dgemm_mfma_ksplit (any D, zero-padded to a multiple of 64; 25.0 us per mat-vec at D = 512, the library GEMM path 33; measurements and the forms
// tried before it -- an LDS-staged kernel, 40-53 us -- in profiles/r04_transfer_mfma.txt): the four waves split
// the INNER dimension, fragments come straight from global memory out of FRAGMENT-PACKED operands -- no LDS and no barrier
// inside the loop; see the comment at the kernel.
// Fragment layout of v_mfma_f64_16x16x4_f64 (cdna_hip_programming.md 3): A[i = lane & 15][k = lane >> 4], B[k = lane >> 4]
// [j = lane & 15], one double each; C/D four doubles per lane: row = (lane >> 4) + 4 reg, col = lane & 15.
// By default D > 512 keeps the rocBLAS path of dsea_krylov.hip.
// Round 5: BOTH products in ONE launch (512 workgroups, T handed over inside the launch by an agent-scope release / acquire
// pair) was built -- results bit-identical to the pair of launches -- and measured at D = 512: 37.6-38.5 us with one workgroup
// per CU, 47.2-48.0 us with K2's workgroups co-resident, against 25.2-26.4 us for the two launches
// (profiles/r05_transfer_single_launch.txt).  The kernel boundary is the cheaper hand-over of 4 MB across eight private L2s.
// Removed.
// -DTFM_DIAG=1/4: timing diagnostics only (no loads in the loop / two k blocks), wrong results.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "dsea_internal.h"

#ifndef TFM_DIAG
#define TFM_DIAG 0
#endif
namespace dsea {

namespace {
typedef double tfm_v4d __attribute__((ext_vector_type(4)));

// Operand convention of the kernel below: C (M x N, ldc) = A' B' with A' (M x K') given as nseg segments of Kseg columns,
// segment s at A + s * segA (row stride lda), and B' either (BT = false) a K' x N row-major matrix in segments B + s * segB
// (row stride ldb), or (BT = true) the TRANSPOSE of an N x K' matrix given the same way (rows j, contiguous along the inner
// index).
// ---- the four waves split the INNER dimension, fragments come straight from global memory ------------------------------------
// Every wave owns the whole TMT x TNT block of 16 x 16 tiles of the workgroup's output tile and every fourth 16-wide block of
// the inner dimension.  No LDS and no barrier inside the loop: the k index of v_mfma_f64_16x16x4_f64 may be permuted (the
// instruction sums over it), so MFMA step s of a block takes k = block + 4 (lane >> 4) + s -- a lane's four A values (and its
// four B values when B' is given transposed) are 32 contiguous bytes of ONE row, two 16-byte loads.  Fragment reads per MFMA:
// (TMT + TNT) / (TMT TNT) instead of 1.5 through LDS; the next block's fragments are requested before this block's MFMAs.  The
// four partial tiles meet once, at the end, through LDS (fixed order: deterministic).
//
// FRAGMENT-PACKED operands (PKA / PKB / PKC).  A lane's 32 bytes sit in a row of the matrix, so one 16-byte load
// instruction of a wave touches 16 rows = 16 cache lines and uses half of each; measured, the fragment loads then take as long as the
// whole kernel (profiles/r04_transfer_mfma.txt).  A packed operand stores every (16-row tile, 16-wide k block) as one 2 KB
// chunk in the order the loads want it: [plane p = 0, 1][lane][2 doubles], lane (i, kq) holding columns 4 kq + 2 p, + 1 of
// row i -- a wave's load instruction is then 1 KB of consecutive bytes.  Chunks of a segment are ordered [tile][k block],
// segments segA / segB doubles apart (a D x D slice packs into exactly D * D doubles).  The tensor's slices are packed once
// at operator creation (k_pack_fragments); K1 WRITES T packed (PKC: its LDS reduction gathers in packed order, every store
// instruction 1 KB of consecutive bytes) for K2 to read.
//
// GUARD (D not a multiple of 64 -- the reference's own examples run D = 20 and D = 80): the packed operands are zero-padded to
// the next multiple of 64, the PLAIN ones are not: a plain B' is read, and a plain C written, only inside `valid` x `valid`
// (reads outside give 0), with scalar accesses (no alignment assumption on an odd leading dimension).
template <int TMT, int TNT, bool BT, bool SWZ, bool PKA, bool PKB, bool PKC, bool GUARD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_dgemm_mfma_ksplit(
    const double* __restrict__ A, int64_t lda, int64_t segA, const double* __restrict__ B, int64_t ldb, int64_t segB,
    double* __restrict__ C, int64_t ldc, int Kseg, int nseg, int valid) {
  constexpr int NW = 4;                                    // waves (8 = two per SIMD measured no faster: profiles/r04_transfer_mfma.txt)
  extern __shared__ __attribute__((aligned(16))) double tfm_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: the block / segment arithmetic stays on the scalar unit)
  const int fi = lane & 15, kq = lane >> 4;
  // tile of this workgroup.  Workgroups go to the 8 XCDs round robin by their linear index and every XCD has its own L2: left
  // as (x, y) = (column tile, row tile), an XCD owns column tiles x = c mod 8 and EVERY row tile, i.e. pulls the whole A'
  // through the fabric.  SWZ gives an XCD a compact (rows / 4) x (columns / 2) block of tiles instead (measured: -2 % on K2,
  // +6 % on K1 whose A' is read once per row tile anyway -- so K2 only).
  int tr = blockIdx.y, tc = blockIdx.x;
  if (SWZ) {
    const int TR = gridDim.y, TC = gridDim.x;
    if ((TR % 4) == 0 && (TC % 2) == 0) {
      const int lin = blockIdx.y * TC + blockIdx.x, xcd = lin & 7, idx = lin >> 3;
      const int br = TR / 4, bc = TC / 2;                  // tiles per XCD block: br x bc (br * bc = TR * TC / 8 = number of idx)
      tr = (xcd >> 1) * br + idx / bc;
      tc = (xcd & 1) * bc + idx % bc;
    }
  }
  const int64_t row0 = (int64_t)tr * (16 * TMT), col0 = (int64_t)tc * (16 * TNT);
  const int bps = Kseg / 16;                               // 16-wide blocks per segment
#if TFM_DIAG == 4                                           /* timing diagnostics: two blocks only = the kernel's fixed cost */
  const int nb = 2;
#else
  const int nb = bps * nseg / NW;                          // blocks of this wave: global block index NW b + wv
#endif
  tfm_v4d acc[TMT][TNT];
#pragma unroll
  for (int tm = 0; tm < TMT; ++tm)
#pragma unroll
    for (int tn = 0; tn < TNT; ++tn) acc[tm][tn] = (tfm_v4d){0.0, 0.0, 0.0, 0.0};
  const double* __restrict__ Arow = A + (row0 + fi) * lda + 4 * kq;
  const double* __restrict__ Bbase = BT ? B + (col0 + fi) * ldb + 4 * kq : B + (int64_t)(4 * kq) * ldb + col0 + fi;
  tfm_v4d fa[TMT], fb[TNT], ga[TMT], gb[TNT];              // current / next block's fragments (element s = MFMA step s)

#define TFM_KS_LOAD(bidx, FA, FB)                                                                                 \
  {                                                                                                               \
    const int gbk_ = NW * (bidx) + wv, sg_ = gbk_ / bps, kb_ = gbk_ - sg_ * bps, kc_ = kb_ * 16;                    \
    const double* __restrict__ ap_ = PKA ? A + (int64_t)sg_ * segA + ((row0 / 16) * bps + kb_) * 256 + 2 * lane  \
                                         : Arow + (int64_t)sg_ * segA + kc_;                                      \
    _Pragma("unroll") for (int tm = 0; tm < TMT; ++tm) {                                                          \
      const double* __restrict__ fa_ = PKA ? ap_ + (int64_t)tm * bps * 256 : ap_ + (int64_t)(16 * tm) * lda;     \
      const double2 lo_ = *reinterpret_cast<const double2*>(fa_);                                                 \
      const double2 hi_ = *reinterpret_cast<const double2*>(fa_ + (PKA ? 128 : 2));                               \
      FA[tm] = (tfm_v4d){lo_.x, lo_.y, hi_.x, hi_.y};                                                             \
    }                                                                                                             \
    if (BT) {                                                                                                     \
      const double* __restrict__ bp_ = PKB ? B + (int64_t)sg_ * segB + ((col0 / 16) * bps + kb_) * 256 + 2 * lane \
                                           : Bbase + (int64_t)sg_ * segB + kc_;                                   \
      _Pragma("unroll") for (int tn = 0; tn < TNT; ++tn) {                                                        \
        const double* __restrict__ fb_ = PKB ? bp_ + (int64_t)tn * bps * 256 : bp_ + (int64_t)(16 * tn) * ldb;   \
        const double2 lo_ = *reinterpret_cast<const double2*>(fb_);                                               \
        const double2 hi_ = *reinterpret_cast<const double2*>(fb_ + (PKB ? 128 : 2));                             \
        FB[tn] = (tfm_v4d){lo_.x, lo_.y, hi_.x, hi_.y};                                                           \
      }                                                                                                           \
    } else {                                                                                                      \
      const double* __restrict__ bp_ = Bbase + (int64_t)sg_ * segB + (int64_t)kc_ * ldb;                          \
      _Pragma("unroll") for (int tn = 0; tn < TNT; ++tn) {                                                        \
        if (GUARD) {                                                                                              \
          const bool cok_ = col0 + fi + 16 * tn < valid;                                                          \
          const int k0_ = kc_ + 4 * kq;                                                                           \
          FB[tn] = (tfm_v4d){cok_ && k0_ < valid ? bp_[16 * tn] : 0.0, cok_ && k0_ + 1 < valid ? bp_[ldb + 16 * tn] : 0.0, \
                             cok_ && k0_ + 2 < valid ? bp_[2 * ldb + 16 * tn] : 0.0,                              \
                             cok_ && k0_ + 3 < valid ? bp_[3 * ldb + 16 * tn] : 0.0};                             \
        } else {                                                                                                  \
          FB[tn] = (tfm_v4d){bp_[16 * tn], bp_[ldb + 16 * tn], bp_[2 * ldb + 16 * tn], bp_[3 * ldb + 16 * tn]};  \
        }                                                                                                         \
      }                                                                                                           \
    }                                                                                                             \
  }
#define TFM_KS_MMA(FA, FB)                                                                                        \
  _Pragma("unroll") for (int s4 = 0; s4 < 4; ++s4)                                                                \
    _Pragma("unroll") for (int tm = 0; tm < TMT; ++tm)                                                            \
      _Pragma("unroll") for (int tn = 0; tn < TNT; ++tn)                                                          \
        acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(FA[tm][s4], FB[tn][s4], acc[tm][tn], 0, 0, 0);

  // an odd number of blocks (small D only): one block on its own first, so that the pipelined loop below always sees pairs
  int b = 0;
  if (nb & 1) {
    TFM_KS_LOAD(0, ga, gb)
    TFM_KS_MMA(ga, gb)
    b = 1;
  }
  if (b < nb) {
    TFM_KS_LOAD(b, fa, fb)
#if TFM_DIAG == 1                                           /* timing diagnostics: no loads inside the loop */
    TFM_KS_LOAD(b + 1, ga, gb)
    for (; b + 2 <= nb; b += 2) {
      TFM_KS_MMA(fa, fb)
      TFM_KS_MMA(ga, gb)
    }
#else
    for (; b + 2 < nb; b += 2) {
      TFM_KS_LOAD(b + 1, ga, gb)
      TFM_KS_MMA(fa, fb)
      TFM_KS_LOAD(b + 2, fa, fb)
      TFM_KS_MMA(ga, gb)
    }
    TFM_KS_LOAD(b + 1, ga, gb)
    TFM_KS_MMA(fa, fb)
    TFM_KS_MMA(ga, gb)
#endif
  }
#undef TFM_KS_LOAD
#undef TFM_KS_MMA

  // the NW waves' partial tiles meet in LDS: [wave][tile][reg r][kq][fi, row padded to 18] (C/D layout: row = kq + 4 r,
  // col = fi).  Wave w then sums tiles w, w + NW, ... over the waves in the order 0, 1, 2, ... and writes them.
  constexpr int NTILE = TMT * TNT;
#pragma unroll
  for (int tm = 0; tm < TMT; ++tm)
#pragma unroll
    for (int tn = 0; tn < TNT; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) tfm_smem[(((wv * NTILE + tm * TNT + tn) * 4 + r) * 4 + kq) * 18 + fi] = acc[tm][tn][r];
  __syncthreads();
#pragma unroll
  for (int t = wv; t < NTILE; t += NW) {
    const int tm = t / TNT, tn = t % TNT;
    if (PKC) {
      // packed store: C is the A-type operand of the next product -- rows are its rows, columns its k.  Chunk of (row tile,
      // k block); position [plane p][lane' = (i, kq')][2]: row i = lane' & 15, columns 4 (lane' >> 4) + 2 p, + 1.
      const int64_t grow = row0 + 16 * tm, seg = grow / Kseg, rt = (grow - seg * Kseg) / 16;     // (C's segments: Kseg rows each)
      const int64_t kb = (col0 + 16 * tn) / 16, cbps = ldc / 16;
      double* __restrict__ cp = C + seg * ((int64_t)Kseg * ldc) + (rt * cbps + kb) * 256 + 2 * lane;
      const int i16 = lane & 15, r = i16 >> 2, q = i16 & 3;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        const int f0 = 4 * (lane >> 4) + 2 * pl;
        double2 v = *reinterpret_cast<const double2*>(&tfm_smem[(((0 * NTILE + t) * 4 + r) * 4 + q) * 18 + f0]);
#pragma unroll
        for (int w2 = 1; w2 < NW; ++w2) {
          const double2 u = *reinterpret_cast<const double2*>(&tfm_smem[(((w2 * NTILE + t) * 4 + r) * 4 + q) * 18 + f0]);
          v.x += u.x;
          v.y += u.y;
        }
        *reinterpret_cast<double2*>(cp + 128 * pl) = v;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double v = tfm_smem[(((0 * NTILE + t) * 4 + r) * 4 + kq) * 18 + fi];
#pragma unroll
        for (int w2 = 1; w2 < NW; ++w2) v += tfm_smem[(((w2 * NTILE + t) * 4 + r) * 4 + kq) * 18 + fi];
        const int64_t crow = row0 + 16 * tm + 4 * r + kq, ccol = col0 + 16 * tn + fi;
        if (!GUARD || (crow < valid && ccol < valid)) C[crow * ldc + ccol] = v;
      }
    }
  }
}

// one-off packing of the tensor's slices (d x D x D row-major) into fragment chunks of the zero-padded d x Dp x Dp tensor
// (Dp = D rounded up to a multiple of 64); one wave per chunk
__global__ __launch_bounds__(256) void k_pack_fragments(const double* __restrict__ B, double* __restrict__ Bp, int D, int Dp,
                                                        int64_t nchunks) {
  const int64_t chunk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (chunk >= nchunks) return;
  const int lane = threadIdx.x & 63, bps = Dp / 16;
  const int64_t per_slice = (int64_t)bps * bps, s = chunk / per_slice, rem = chunk - s * per_slice;
  const int64_t rt = rem / bps, kb = rem - rt * bps;
  const int64_t row = rt * 16 + (lane & 15), c0 = kb * 16 + 4 * (lane >> 4);
  const double* __restrict__ src = B + s * (int64_t)D * D + row * (int64_t)D + c0;
  double v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = (row < D && c0 + e < D) ? src[e] : 0.0;
  double* __restrict__ dst = Bp + chunk * 256 + 2 * lane;
  *reinterpret_cast<double2*>(dst) = make_double2(v[0], v[1]);
  *reinterpret_cast<double2*>(dst + 128) = make_double2(v[2], v[3]);
}

}  // namespace

bool transfer_mfma_applicable(const OpDesc& op) {
  return op.kind == OP_TRANSFER && op.transfer.D >= 1 && op.transfer.d >= 1 && op.transfer.Bp && op.transfer.Tp;
}

namespace {
template <int TMT, int TNT, bool BT, bool SWZ, bool PKA, bool PKB, bool PKC, bool GUARD>
int tfm_launch_ksplit(dim3 grid, hipStream_t st, const double* A, int64_t lda, int64_t sA, const double* B, int64_t ldb,
                      int64_t sB, double* C, int64_t ldc, int Kseg, int nseg, int valid) {
  constexpr size_t lds = (size_t)4 * TMT * TNT * 4 * 4 * 18 * sizeof(double);
  if (lds > 65536) {
    static thread_local int attr_dev = -1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (dev != attr_dev) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_dgemm_mfma_ksplit<TMT, TNT, BT, SWZ, PKA, PKB, PKC, GUARD>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return -1;
      attr_dev = dev;
    }
  }
  hipLaunchKernelGGL((k_dgemm_mfma_ksplit<TMT, TNT, BT, SWZ, PKA, PKB, PKC, GUARD>), grid, dim3(256), lds, st, A, lda, sA, B, ldb,
                     sB, C, ldc, Kseg, nseg, valid);
  return 0;
}
}  // namespace

// fragment-packed, zero-padded copy of the slices (dsea_op_create_transfer, once): Bp must hold d * Dp * Dp doubles
void launch_pack_fragments(const double* B, double* Bp, int D, int d, hipStream_t st) {
  const int Dp = (D + 63) / 64 * 64;
  const int64_t nchunks = (int64_t)d * (Dp / 16) * (Dp / 16);
  hipLaunchKernelGGL(k_pack_fragments, dim3((unsigned)((nchunks + 3) / 4)), dim3(256), 0, st, B, Bp, D, Dp, nchunks);
}

// y = sum_s B_s X B_s^T through two launches; Tp = the operator's d x Dp x Dp scratch (written and read in packed order), Bp =
// the fragment-packed slices, Dp = D rounded up to a multiple of 64.  Returns 0 or -1 (not applicable).
int launch_transfer_mfma(const OpDesc& op, const double* x, double* y, hipStream_t st) {
  if (!transfer_mfma_applicable(op)) return -1;
  const TransferParams& p = op.transfer;
  const int D = p.D, d = p.d, Dp = (D + 63) / 64 * 64;
  const int64_t DDp = (int64_t)Dp * Dp;
  const bool guard = Dp != D;
  const char* env = getenv("DSEA_TRANSFER_TILE");          // (A/B measurements: "s" / "l" force the small / large tiles)
  // 64 x 64 tiles for both products (0.5 fragment reads per MFMA instead of 0.75 / 1) where their workgroup count wastes no
  // more of its last round of 256 CUs than the small tiles' does -- the rule that matches every size measured
  // (profiles/r04_transfer_mfma.txt: D = 896, 1024, 2048 large; 512, 640, 768, 1280, 1536 small)
  const auto waste = [](int64_t c) { return (double)((c + 255) / 256 * 256 - c) / (double)c; };
  const int64_t ns = (int64_t)(Dp / 32) * (Dp / 32), nl = (int64_t)(Dp / 64) * (Dp / 64);
  const bool large = env ? env[0] == 'l' : waste(nl) <= waste(ns);
  if (large) {
    const dim3 h1((unsigned)(Dp / 64), (unsigned)((int64_t)d * Dp / 64)), h2((unsigned)(Dp / 64), (unsigned)(Dp / 64));
    int rc = guard ? tfm_launch_ksplit<4, 4, false, true, true, false, true, true>(h1, st, p.Bp, Dp, 0, x, D, 0, p.Tp, Dp, Dp, 1, D)
                   : tfm_launch_ksplit<4, 4, false, true, true, false, true, false>(h1, st, p.Bp, Dp, 0, x, D, 0, p.Tp, Dp, Dp, 1, D);
    if (rc != 0) return rc;
    return guard ? tfm_launch_ksplit<4, 4, true, true, true, true, false, true>(h2, st, p.Tp, Dp, DDp, p.Bp, Dp, DDp, y, D, Dp, d, D)
                 : tfm_launch_ksplit<4, 4, true, true, true, true, false, false>(h2, st, p.Tp, Dp, DDp, p.Bp, Dp, DDp, y, D, Dp, d, D);
  }
  // K1: T (d Dp x Dp) = B (the d padded slices stacked) X (D x D, read as Dp x Dp with zeros outside): tile 64 x 32; reads the
  // packed slices, writes T packed
  const dim3 g1((unsigned)(Dp / 32), (unsigned)((int64_t)d * Dp / 64));
  int rc = guard ? tfm_launch_ksplit<4, 2, false, false, true, false, true, true>(g1, st, p.Bp, Dp, 0, x, D, 0, p.Tp, Dp, Dp, 1, D)
                 : tfm_launch_ksplit<4, 2, false, false, true, false, true, false>(g1, st, p.Bp, Dp, 0, x, D, 0, p.Tp, Dp, Dp, 1, D);
  if (rc != 0) return rc;
  // K2: y (D x D) = sum_s T_s B_s^T: inner dimension in d segments of Dp; tile 32 x 32; reads T and the slices packed
  const dim3 g2((unsigned)(Dp / 32), (unsigned)(Dp / 32));
  return guard ? tfm_launch_ksplit<2, 2, true, true, true, true, false, true>(g2, st, p.Tp, Dp, DDp, p.Bp, Dp, DDp, y, D, Dp, d, D)
               : tfm_launch_ksplit<2, 2, true, true, true, true, false, false>(g2, st, p.Tp, Dp, DDp, p.Bp, Dp, DDp, y, D, Dp, d, D);
}

}  // namespace dsea
