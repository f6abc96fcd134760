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
// 64 x 32 (K1: 16 x 16 = 256 workgroups) / 32 x 32 (K2: 256 workgroups) tile -- one workgroup per CU, one wave per SIMD,
// each wave 16 x 32 / 16 x 16 of it as 16 x 16 x 4 MFMA tiles.  Inner dimension in chunks of KC staged through LDS (double
// buffered: the next chunk's global loads are in flight while the current one is multiplied; one barrier per chunk).
// LDS rows are padded so that the 8-byte fragment reads are conflict-free: an A-type tile [rows][KC + 2] (lane (i, kk)
// reads word pair 2 i + kk mod 32 ... distinct over the 32 lanes of an LDS cycle), a B tile [KC][TN + 16].
// Fragment layout of v_mfma_f64_16x16x4_f64 (cdna_hip_programming.md 3): A[i = lane & 15][k = lane >> 4], B[k = lane >> 4]
// [j = lane & 15], one double each; C/D four doubles per lane: row = (lane >> 4) + 4 reg, col = lane & 15.
// Shapes the kernels do not cover (D not a multiple of 64) keep the rocBLAS path of dsea_krylov.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dsea_internal.h"

#ifndef TFM_DIAG
#define TFM_DIAG 0
#endif
namespace dsea {

namespace {
typedef double tfm_v4d __attribute__((ext_vector_type(4)));

// C (M x N, ldc) = A' B' with A' (M x K') given as nseg segments of Kseg columns, segment s at A + s * segA (row stride lda),
// and B' either (BT = false) a K' x N row-major matrix in segments B + s * segB (row stride ldb), or (BT = true) the
// TRANSPOSE of an N x K' matrix given the same way (rows j, contiguous along the inner index).
template <int TM, int TN, int KC, int WM, int WN, bool BT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_dgemm_mfma(const double* __restrict__ A, int64_t lda, int64_t segA,
                                                    const double* __restrict__ B, int64_t ldb, int64_t segB,
                                                    double* __restrict__ C, int64_t ldc, int Kseg, int nseg) {
  static_assert(WM * WN == 4, "four waves");
  constexpr int WTM = TM / WM, WTN = TN / WN;            // the wave's part of the tile
  static_assert(WTM == 16, "one MFMA tile row per wave");
  constexpr int NT = WTN / 16;                            // MFMA tiles along N per wave
  constexpr int SA = KC + 2;                              // padded row of an A-type tile (doubles)
  constexpr int SB = BT ? KC + 2 : TN + 16;               // B tile: [TN][KC + 2] or [KC][TN + 16]
  constexpr int A_ELEMS = TM * SA, B_ELEMS = BT ? TN * SB : KC * SB;
  extern __shared__ __attribute__((aligned(16))) double tfm_smem[];
  double* As = tfm_smem;                                  // [2][A_ELEMS]
  double* Bs = tfm_smem + 2 * A_ELEMS;                    // [2][B_ELEMS]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv / WN, wn = wv % WN;
  const int64_t row0 = (int64_t)blockIdx.y * TM, col0 = (int64_t)blockIdx.x * TN;
  // global -> register staging: A tile = TM rows x KC doubles, as double2
  constexpr int A_V2 = TM * KC / 2 / 256;                 // double2 per thread
  constexpr int B_V2 = (BT ? TN * KC : KC * TN) / 2 / 256;
  static_assert(A_V2 >= 1 && B_V2 >= 1, "tile too small for 256 threads");
  double2 ra[A_V2], rb[B_V2];
  const int chunks_per_seg = Kseg / KC, nchunks = chunks_per_seg * nseg;

  // (macros, not lambdas: a by-reference capture of the staging arrays leaves them in scratch memory)
#define TFM_LOAD_CHUNK(cidx)                                                                                   \
  {                                                                                                            \
    const int s_ = (cidx) / chunks_per_seg, k0_ = ((cidx)-s_ * chunks_per_seg) * KC;                            \
    const double* __restrict__ Ab_ = A + (int64_t)s_ * segA + row0 * lda + k0_;                                \
    _Pragma("unroll") for (int p_ = 0; p_ < A_V2; ++p_) {                                                      \
      const int idx_ = tid + 256 * p_, r_ = idx_ / (KC / 2), c2_ = idx_ % (KC / 2);                             \
      ra[p_] = *reinterpret_cast<const double2*>(Ab_ + (int64_t)r_ * lda + 2 * c2_);                           \
    }                                                                                                          \
    if (BT) {                                                                                                  \
      const double* __restrict__ Bb_ = B + (int64_t)s_ * segB + col0 * ldb + k0_;                              \
      _Pragma("unroll") for (int p_ = 0; p_ < B_V2; ++p_) {                                                    \
        const int idx_ = tid + 256 * p_, r_ = idx_ / (KC / 2), c2_ = idx_ % (KC / 2);                           \
        rb[p_] = *reinterpret_cast<const double2*>(Bb_ + (int64_t)r_ * ldb + 2 * c2_);                         \
      }                                                                                                        \
    } else {                                                                                                   \
      const double* __restrict__ Bb_ = B + (int64_t)s_ * segB + (int64_t)k0_ * ldb + col0;                     \
      _Pragma("unroll") for (int p_ = 0; p_ < B_V2; ++p_) {                                                    \
        const int idx_ = tid + 256 * p_, r_ = idx_ / (TN / 2), c2_ = idx_ % (TN / 2);                           \
        rb[p_] = *reinterpret_cast<const double2*>(Bb_ + (int64_t)r_ * ldb + 2 * c2_);                         \
      }                                                                                                        \
    }                                                                                                          \
  }
#define TFM_STORE_CHUNK(bufidx)                                                                                \
  {                                                                                                            \
    double* Ad_ = As + (bufidx)*A_ELEMS;                                                                       \
    double* Bd_ = Bs + (bufidx)*B_ELEMS;                                                                       \
    _Pragma("unroll") for (int p_ = 0; p_ < A_V2; ++p_) {                                                      \
      const int idx_ = tid + 256 * p_, r_ = idx_ / (KC / 2), c2_ = idx_ % (KC / 2);                             \
      *reinterpret_cast<double2*>(Ad_ + r_ * SA + 2 * c2_) = ra[p_];                                           \
    }                                                                                                          \
    _Pragma("unroll") for (int p_ = 0; p_ < B_V2; ++p_) {                                                      \
      const int idx_ = tid + 256 * p_;                                                                         \
      const int r_ = BT ? idx_ / (KC / 2) : idx_ / (TN / 2), c2_ = BT ? idx_ % (KC / 2) : idx_ % (TN / 2);      \
      *reinterpret_cast<double2*>(Bd_ + r_ * SB + 2 * c2_) = rb[p_];                                           \
    }                                                                                                          \
  }

  tfm_v4d acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (tfm_v4d){0.0, 0.0, 0.0, 0.0};
  const int fi = lane & 15, fk = lane >> 4;

  TFM_LOAD_CHUNK(0)
  TFM_STORE_CHUNK(0)
  __syncthreads();
#define TFM_COMPUTE(bufidx)                                                                                     \
  {                                                                                                             \
    const double* __restrict__ Ad = As + (bufidx)*A_ELEMS + (wm * 16 + fi) * SA + fk;                           \
    const double* __restrict__ Bd = Bs + (bufidx)*B_ELEMS;                                                      \
    _Pragma("unroll") for (int ks = 0; ks < KC; ks += 4) {                                                      \
      const double av = Ad[ks];                                                                                 \
      _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                          \
        const int j = wn * WTN + t * 16 + fi;                                                                   \
        const double bv = BT ? Bd[j * SB + ks + fk] : Bd[(ks + fk) * SB + j];                                   \
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[t], 0, 0, 0);                                 \
      }                                                                                                         \
    }                                                                                                           \
  }
  for (int c = 0; c + 1 < nchunks; ++c) {
    const int buf = c & 1;
#if TFM_DIAG != 1
    TFM_LOAD_CHUNK(c + 1)                                  // in flight while this chunk is multiplied
#endif
#if TFM_DIAG != 2
    TFM_COMPUTE(buf)
#endif
    // Pin the order loads -> MFMAs -> LDS stores: left alone, the scheduler either sinks the loads next to their stores
    // (saving registers) or hoists the stores above the MFMAs -- both expose the load latency once per chunk.  The empty
    // asm makes every staged value depend on the last accumulator of the chunk.
    {
      const double dep_ = acc[NT - 1][3];
#pragma unroll
      for (int p_ = 0; p_ < A_V2; ++p_) asm volatile("" : "+v"(ra[p_].x), "+v"(ra[p_].y) : "v"(dep_));
#pragma unroll
      for (int p_ = 0; p_ < B_V2; ++p_) asm volatile("" : "+v"(rb[p_].x), "+v"(rb[p_].y) : "v"(dep_));
    }
    TFM_STORE_CHUNK(buf ^ 1)                               // (last read in iteration c - 1: every wave is past that barrier)
    __syncthreads();
  }
  TFM_COMPUTE((nchunks - 1) & 1)
#undef TFM_COMPUTE
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int64_t col = col0 + wn * WTN + t * 16 + fi;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = row0 + wm * 16 + 4 * r + fk;
      C[row * ldc + col] = acc[t][r];
    }
  }
#undef TFM_LOAD_CHUNK
#undef TFM_STORE_CHUNK
}

template <int TM, int TN, int KC, bool BT>
constexpr size_t tfm_lds_bytes() {
  return (size_t)2 * ((size_t)TM * (KC + 2) + (BT ? (size_t)TN * (KC + 2) : (size_t)KC * (TN + 16))) * sizeof(double);
}
}  // namespace

bool transfer_mfma_applicable(const OpDesc& op) {
  return op.kind == OP_TRANSFER && op.transfer.D >= 64 && (op.transfer.D % 64) == 0 && op.transfer.d >= 1;
}

namespace {
// one GEMM launch of the pair; the dynamic-LDS attribute (above the 64 KB a kernel gets without asking) is set once per
// device and instantiation
template <int TM, int TN, int KC, int WM, int WN, bool BT>
int tfm_launch(dim3 grid, hipStream_t st, const double* A, int64_t lda, int64_t sA, const double* B, int64_t ldb, int64_t sB,
               double* C, int64_t ldc, int Kseg, int nseg) {
  static thread_local int attr_dev = -1;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (dev != attr_dev) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_dgemm_mfma<TM, TN, KC, WM, WN, BT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)tfm_lds_bytes<TM, TN, KC, BT>()) != hipSuccess)
      return -1;
    attr_dev = dev;
  }
  hipLaunchKernelGGL((k_dgemm_mfma<TM, TN, KC, WM, WN, BT>), grid, dim3(256), (tfm_lds_bytes<TM, TN, KC, BT>()), st, A, lda, sA,
                     B, ldb, sB, C, ldc, Kseg, nseg);
  return 0;
}
}  // namespace

// y = sum_s B_s X B_s^T through the two kernels above; T = the operator's d x D x D scratch.  Returns 0 or -1 (not applicable).
// The chunk of the inner dimension is as long as D allows (D % 128 == 0: 64 for K1, 128 for K2 -- 32 MFMAs per wave and
// chunk, ~1 us, which covers the latency of the next chunk's global loads; otherwise half that).
int launch_transfer_mfma(const OpDesc& op, const double* x, double* y, hipStream_t st) {
  if (!transfer_mfma_applicable(op)) return -1;
  const TransferParams& p = op.transfer;
  const int D = p.D, d = p.d;
  const int64_t DD = (int64_t)D * D;
  const bool longc = (D % 128) == 0;
  // K1: T (dD x D) = B (dD x D, the d slices stacked) X (D x D): tile 64 x 32, waves 4 x 1
  const dim3 g1((unsigned)(D / 32), (unsigned)((int64_t)d * D / 64));
  int rc = longc ? tfm_launch<64, 32, 64, 4, 1, false>(g1, st, p.B, D, 0, x, D, 0, p.T, D, D, 1)
                 : tfm_launch<64, 32, 32, 4, 1, false>(g1, st, p.B, D, 0, x, D, 0, p.T, D, D, 1);
  if (rc != 0) return rc;
  // K2: y (D x D) = sum_s T_s B_s^T: inner dimension in d segments of D; tile 32 x 32, waves 2 x 2
  const dim3 g2((unsigned)(D / 32), (unsigned)(D / 32));
  return longc ? tfm_launch<32, 32, 128, 2, 2, true>(g2, st, p.T, D, DD, p.B, D, DD, y, D, D, d)
               : tfm_launch<32, 32, 64, 2, 2, true>(g2, st, p.T, D, DD, p.B, D, DD, y, D, D, d);
}

}  // namespace dsea
