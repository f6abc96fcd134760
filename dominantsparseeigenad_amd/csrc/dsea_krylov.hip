// dsea_krylov.hip -- the NON-symmetric side of the hot path (BASELINE config 4):
//   * dense and MPS-transfer-matrix operands whose mat-vec is GEMM-shaped and goes to rocBLAS (the one place of
//     this library where a matrix core is the right unit: reference examples/TFIM_vumps/general.py:59-66 applies
//     sum_s A_s r A_s^T as einsums on the host);
//   * the Arnoldi factorisation loop and the restarted-GMRES cycle ON THE DEVICE, without a host round trip per
//     step: what reference eig.py:29-30,54-57,116-117,137-144 delegates to SciPy's ARPACK `eigs` / `gmres`.
// The orthogonalisation reuses the basis-streaming kernels of the Lanczos path (dots pass + correction pass,
// classical Gram-Schmidt); the second pass runs only when the DGKS test asks for it (ARPACK's rule), decided on
// the device through the same "skip record" the Lanczos breakdown logic uses.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>

#include "dsea_internal.h"
#include "dsea_device.h"

namespace dsea {

// ------------------------------------------------------------------------------------------
// rocBLAS, bound at run time: the process already holds the copy PyTorch-ROCm loaded; binding by dlopen keeps
// libdsea.so free of a link-time dependency on one particular rocBLAS build.
// ------------------------------------------------------------------------------------------
namespace {
struct Blas {
  bool ok = false;
  rocblas_handle h = nullptr;
  decltype(&rocblas_create_handle) create = nullptr;
  decltype(&rocblas_destroy_handle) destroy = nullptr;
  decltype(&rocblas_set_stream) set_stream = nullptr;
  decltype(&rocblas_set_pointer_mode) set_pointer_mode = nullptr;
  decltype(&rocblas_set_atomics_mode) set_atomics_mode = nullptr;
  decltype(&rocblas_dgemv) dgemv = nullptr;
  decltype(&rocblas_dgemm) dgemm = nullptr;
  decltype(&rocblas_dgemm_strided_batched) dgemm_sb = nullptr;
};
Blas g_blas;
std::once_flag g_blas_once;
// One rocBLAS handle PER (DEVICE, STREAM) (created on first use, bound to its stream once): switching the stream of a
// shared handle while its earlier work may still be running is not safe for kernels that use the handle's device
// workspace, and the left / right solves of the non-symmetric primitives run concurrently on two streams
// (eig._two_sides).  The device is part of the key because PyTorch's default stream is the null stream on EVERY device:
// a handle (and its device workspace) created on cuda:0 must not serve cuda:1.  The table holds MAX_STREAM_HANDLES
// entries; beyond that the least recently used handle is destroyed and its slot reused (PyTorch's stream pool alone
// has 32 streams), so the table never refuses a stream.
constexpr int MAX_STREAM_HANDLES = 16;
struct HandleSlot {
  int device;
  hipStream_t stream;
  rocblas_handle handle;
  uint64_t last_use;
};
HandleSlot g_slots[MAX_STREAM_HANDLES];
int g_handles = 0;
uint64_t g_use_clock = 0;

void blas_init() {
  void* lib = dlopen("librocblas.so.5", RTLD_NOW | RTLD_NOLOAD);
  if (!lib) lib = dlopen("librocblas.so", RTLD_NOW | RTLD_NOLOAD);
  if (!lib) lib = dlopen("librocblas.so.5", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) lib = dlopen("librocblas.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) return;
#define BIND(field, name)                                                   \
  g_blas.field = reinterpret_cast<decltype(g_blas.field)>(dlsym(lib, name)); \
  if (!g_blas.field) return;
  BIND(create, "rocblas_create_handle")
  BIND(destroy, "rocblas_destroy_handle")
  BIND(set_stream, "rocblas_set_stream")
  BIND(set_pointer_mode, "rocblas_set_pointer_mode")
  BIND(set_atomics_mode, "rocblas_set_atomics_mode")
  BIND(dgemv, "rocblas_dgemv")
  BIND(dgemm, "rocblas_dgemm")
  BIND(dgemm_sb, "rocblas_dgemm_strided_batched")
#undef BIND
  if (g_blas.create(&g_blas.h) != rocblas_status_success) return;   // probe: the library is usable
  g_blas.destroy(g_blas.h);   // handles are per (device, stream): created on first use in handle_for
  g_blas.h = nullptr;
  g_blas.ok = true;
}
std::mutex g_blas_mutex;  // guards the handle table (calls on distinct handles run concurrently)

// the handle of (current device, `st`) (nullptr on failure)
rocblas_handle handle_for(hipStream_t st) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(g_blas_mutex);
  ++g_use_clock;
  for (int i = 0; i < g_handles; ++i)
    if (g_slots[i].device == dev && g_slots[i].stream == st) {
      g_slots[i].last_use = g_use_clock;
      return g_slots[i].handle;
    }
  int slot = g_handles;
  if (g_handles >= MAX_STREAM_HANDLES) {   // evict the least recently used handle (its work is stream-ordered: the
    slot = 0;                              // destroy call synchronises that handle's stream inside rocBLAS)
    for (int i = 1; i < g_handles; ++i)
      if (g_slots[i].last_use < g_slots[slot].last_use) slot = i;
    int cur = dev;
    if (g_slots[slot].device != cur) (void)hipSetDevice(g_slots[slot].device);
    g_blas.destroy(g_slots[slot].handle);
    if (g_slots[slot].device != cur) (void)hipSetDevice(cur);
    g_slots[slot].handle = nullptr;
  }
  rocblas_handle h = nullptr;
  if (g_blas.create(&h) != rocblas_status_success) {
    if (slot < g_handles) {   // keep the table dense: move the last entry into the freed slot
      g_slots[slot] = g_slots[g_handles - 1];
      --g_handles;
    }
    return nullptr;
  }
  if (g_blas.set_stream(h, st) != rocblas_status_success) {
    g_blas.destroy(h);
    if (slot < g_handles) {
      g_slots[slot] = g_slots[g_handles - 1];
      --g_handles;
    }
    return nullptr;
  }
  g_blas.set_pointer_mode(h, rocblas_pointer_mode_host);
  g_blas.set_atomics_mode(h, rocblas_atomics_not_allowed);  // bit-repeatable runs, like the rest of the path
  g_slots[slot] = HandleSlot{dev, st, h, g_use_clock};
  if (slot == g_handles) ++g_handles;
  return h;
}
}  // namespace

bool blas_available() {
  std::call_once(g_blas_once, blas_init);
  return g_blas.ok;
}

// out[b] = in[b]^T for `batch` square D x D row-major matrices (32 x 32 tiles through LDS, coalesced both ways)
__global__ __launch_bounds__(256) void k_transpose_sq(const double* __restrict__ in, double* __restrict__ out, int D) {
  __shared__ double tile[32][33];
  const int64_t base = (int64_t)blockIdx.z * D * D;
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = by + ty + 8 * k, c = bx + tx;
    tile[ty + 8 * k][tx] = (r < D && c < D) ? in[base + (int64_t)r * D + c] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = bx + ty + 8 * k, c = by + tx;
    if (r < D && c < D) out[base + (int64_t)r * D + c] = tile[tx][ty + 8 * k];
  }
}

void launch_transpose_sq(const double* in, double* out, int D, int batch, hipStream_t st) {
  const unsigned g = (unsigned)((D + 31) / 32);
  hipLaunchKernelGGL(k_transpose_sq, dim3(g, g, (unsigned)batch), dim3(256), 0, st, in, out, D);
}

// y = Y[0] + Y[1] + ... + Y[d-1]   (fixed order)
__global__ __launch_bounds__(256) void k_sum_slices(const double* __restrict__ Y, int d, int64_t n,
                                                    double* __restrict__ y) {
  const int64_t stride = (int64_t)gridDim.x * 512;
  for (int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2; row < n; row += stride) {
    double2 acc = ld2<true>(Y, row, n);
    for (int k = 1; k < d; ++k) {
      const double2 v = ld2<true>(Y + (int64_t)k * n, row, n);
      acc.x += v.x;
      acc.y += v.y;
    }
    st2<true>(y, row, n, acc);
  }
}

// ------------------------------------------------------------------------------------------
// general dense operand (reference eig.py:28-30): hand-written GEMV, HBM-bound (n^2 doubles read once), deterministic
// ------------------------------------------------------------------------------------------
// y = A x, A row-major: a wave takes ROWS rows at a time, its lanes stride along the columns -- 16 bytes per lane and row
// when everything is pair-aligned (VEC), scalar accesses otherwise (odd n or lda: the small matrices of the reference's
// tests); x is read once per ROWS rows.  ROWS = 1 with two column steps in flight up to n = 8192 (n waves: enough of
// them to cover the latency), 4 beyond (x re-read four times less).  Fixed summation order per row: two lane-strided fma
// chains (even / odd column steps), their sum, then wave_sum.
template <bool VEC, int ROWS>
__global__ __launch_bounds__(256) void k_gemv_rows(const double* __restrict__ A, int64_t lda, int64_t n,
                                                   const double* __restrict__ x, double* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  for (int64_t r0 = wave * ROWS; r0 < n; r0 += (int64_t)gridDim.x * 4 * ROWS) {
    double acc[ROWS], acc2[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = acc2[r] = 0.0;
    const double* __restrict__ a0 = A + r0 * lda;
    if (VEC) {
      int64_t c = 2 * lane;
      for (; c + 128 < n; c += 256) {
        const double2 x0 = *reinterpret_cast<const double2*>(x + c), x1 = *reinterpret_cast<const double2*>(x + c + 128);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          if (r0 + r < n) {
            const double2 a0v = ld2_stream<false>(a0 + r * lda, c, n);          // (read once: non-temporal)
            const double2 a1v = ld2_stream<false>(a0 + r * lda, c + 128, n);
            acc[r] = fma(a0v.x, x0.x, acc[r]);
            acc[r] = fma(a0v.y, x0.y, acc[r]);
            acc2[r] = fma(a1v.x, x1.x, acc2[r]);
            acc2[r] = fma(a1v.y, x1.y, acc2[r]);
          }
        }
      }
      if (c < n) {
        const double2 x0 = *reinterpret_cast<const double2*>(x + c);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          if (r0 + r < n) {
            const double2 a0v = ld2_stream<false>(a0 + r * lda, c, n);
            acc[r] = fma(a0v.x, x0.x, acc[r]);
            acc[r] = fma(a0v.y, x0.y, acc[r]);
          }
        }
      }
    } else {
      for (int64_t c = lane; c < n; c += 64) {
        const double xv = x[c];
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
          if (r0 + r < n) acc[r] = fma(a0[r * lda + c], xv, acc[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const double t = wave_sum(acc[r] + acc2[r]);
      if (lane == 0 && r0 + r < n) y[r0 + r] = t;
    }
  }
}

// y = A^T x without a transposed copy and without scratch: a block owns a strip of 64 columns, its four waves take every
// fourth row (512 contiguous bytes per row and wave), the four partial strips meet in LDS in fixed order.  n / 64 blocks:
// correct and deterministic, but it fills the chip only from n ~ 16 384 -- callers that apply A^T repeatedly hand in the
// transposed matrix instead (operators.DenseOperator does).
__global__ __launch_bounds__(256) void k_gemv_cols(const double* __restrict__ A, int64_t lda, int64_t n,
                                                   const double* __restrict__ x, double* __restrict__ y) {
  __shared__ double part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int64_t c0 = (int64_t)blockIdx.x * 64; c0 < n; c0 += (int64_t)gridDim.x * 64) {
    const int64_t c = c0 + lane;
    double acc = 0.0;
    if (c < n)
      for (int64_t r = w; r < n; r += 4) acc = fma(A[r * lda + c], x[r], acc);
    part[w][lane] = acc;
    __syncthreads();
    if (w == 0 && c < n) y[c] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
    __syncthreads();
  }
}

int launch_gemv(const DenseParams& p, const double* x, double* y, hipStream_t st) {
  if (p.transpose) {
    int64_t nb = (p.n + 63) / 64;
    if (nb > 65535) nb = 65535;
    hipLaunchKernelGGL(k_gemv_cols, dim3((unsigned)nb), dim3(256), 0, st, p.A, p.lda, p.n, x, y);
    return 0;
  }
  const bool vec = (p.n % 2) == 0 && (p.lda % 2) == 0 && (reinterpret_cast<uintptr_t>(p.A) & 15u) == 0 &&
                   (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
  if (p.n <= 8192) {
    const int64_t nb = (p.n + 3) / 4;                       // one row per wave
    if (vec) hipLaunchKernelGGL((k_gemv_rows<true, 1>), dim3((unsigned)nb), dim3(256), 0, st, p.A, p.lda, p.n, x, y);
    else hipLaunchKernelGGL((k_gemv_rows<false, 1>), dim3((unsigned)nb), dim3(256), 0, st, p.A, p.lda, p.n, x, y);
  } else {
    int64_t nb = (p.n + 15) / 16;
    if (nb > 4096) nb = 4096;
    if (vec) hipLaunchKernelGGL((k_gemv_rows<true, 4>), dim3((unsigned)nb), dim3(256), 0, st, p.A, p.lda, p.n, x, y);
    else hipLaunchKernelGGL((k_gemv_rows<false, 4>), dim3((unsigned)nb), dim3(256), 0, st, p.A, p.lda, p.n, x, y);
  }
  return 0;
}

// y = op x for the GEMM-shaped operands (row-major data, rocBLAS is column-major: a row-major product C = A B
// is the column-major product C^T = B^T A^T on the same memory).  Returns 0 or -1.
int blas_apply(const OpDesc& op, const double* x, double* y, hipStream_t st) {
  // the transfer operand has two hand-written fp64 MFMA kernels (dsea_transfer_mfma.hip; any D, zero-padded to a multiple of 64).
  // They are the default where they are measured faster than the library GEMMs below (profiles/r04_transfer_mfma.txt: up to
  // D = 768 -- 9-89 us against 20-110; within +-3 % of each other beyond, up to 2048) and the only path where rocBLAS is absent.
  // DSEA_TRANSFER_MFMA=0 -> library GEMMs always; =1 -> the hand-written kernels wherever they apply.
  if (op.kind == OP_TRANSFER) {
    const char* env = getenv("DSEA_TRANSFER_MFMA");
    const int D = op.transfer.D;
    const bool by_size = D <= 768;
    const bool want = env ? env[0] != '0' : by_size;
    if ((want || !blas_available()) && launch_transfer_mfma(op, x, y, st) == 0) return 0;
  }
  // the general dense operand: hand-written GEMV (DSEA_DENSE_GEMV=0 -> rocBLAS, for A/B measurements)
  if (op.kind == OP_DENSE) {
    const char* env = getenv("DSEA_DENSE_GEMV");
    if (!(env && env[0] == '0' && blas_available())) return launch_gemv(op.dense, x, y, st);
  }
  if (!blas_available()) return -1;
  const double one = 1.0, zero = 0.0;
  rocblas_handle hd = handle_for(st);
  if (!hd) return -1;
  rocblas_status rs = rocblas_status_success;
  if (op.kind == OP_DENSE) {
    const DenseParams& p = op.dense;
    // row-major A (n x n, lda) is the column-major A^T: y = A x = op_T(mem) x ; y = A^T x = op_N(mem) x
    rs = g_blas.dgemv(hd, p.transpose ? rocblas_operation_none : rocblas_operation_transpose, (rocblas_int)p.n,
                      (rocblas_int)p.n, &one, p.A, (rocblas_int)p.lda, x, 1, &zero, y, 1);
  } else if (op.kind == OP_TRANSFER) {
    // y = sum_k B_k x B_k^T with B = A (general.py:59-61 "fr") or B_k = A_k^T (general.py:62-64 "fl": the same form on
    // the transposed tensor, copied once at creation).  Both contractions are issued in the row-major "X Y^T" shape,
    // the one rocBLAS runs fastest for 512^3 fp64 (measured: X Y^T 12.6 us per pair of 512^3 products, X Y 18.8 us,
    // one (D x dD)(dD x D) product 35 us): T_k = B_k (x^T)^T with x^T from a small transpose kernel, Y_k = T_k B_k^T,
    // then y = sum_k Y_k in fixed order.  Row-major C = X Y^T is the column-major product C^T = Y X^T: gemm(T, N, ...)
    // with Y's memory first.
    const TransferParams& p = op.transfer;
    const rocblas_int D = p.D;
    const rocblas_stride DD = (rocblas_stride)p.D * p.D;
    launch_transpose_sq(x, p.xT, p.D, 1, st);
    rs = g_blas.dgemm_sb(hd, rocblas_operation_transpose, rocblas_operation_none, D, D, D, &one, p.xT, D, 0, p.B, D,
                         DD, &zero, p.T, D, DD, p.d);
    if (rs == rocblas_status_success)
      rs = g_blas.dgemm_sb(hd, rocblas_operation_transpose, rocblas_operation_none, D, D, D, &one, p.B, D, DD, p.T,
                           D, DD, &zero, p.Y, D, DD, p.d);
    if (rs == rocblas_status_success) {
      int64_t nb = ((int64_t)DD + 511) / 512;
      if (nb > 2048) nb = 2048;
      hipLaunchKernelGGL(k_sum_slices, dim3((unsigned)nb), dim3(256), 0, st, (const double*)p.Y, p.d, (int64_t)DD, y);
    }
  } else {
    return -1;
  }
  return rs == rocblas_status_success ? 0 : -1;
}

// ------------------------------------------------------------------------------------------
// Arnoldi step scalars
// ------------------------------------------------------------------------------------------
// DGKS test after the first Gram-Schmidt pass: a second pass is needed iff ||w - V V^T w||^2 < 1/2 ||w||^2.
// skip[0] = 1 -> the second pass kernels return at once.  (c1[i] = ||w||^2 from the dots pass.)
// (The second stage of ||w1||^2 -- the sum of the correction pass's `count` partials, in the order of
//  k_cg_finalize_slot -- is done here as well: one launch instead of two small dependent ones per step.)
__global__ __launch_bounds__(256) void k_dgks_decide(const double* __restrict__ c1, int i,
                                                     const double* __restrict__ P, int count,
                                                     double* __restrict__ nrm1, double* __restrict__ skip,
                                                     const double* __restrict__ brk, double* __restrict__ counter) {
  __shared__ double sm4[4];
  if (broken(brk)) {
    if (threadIdx.x == 0) skip[0] = 1.0;
    return;
  }
  double acc = 0.0;
  for (int b = threadIdx.x; b < count; b += 256) acc += P[b];
  const double t = block_sum(acc, sm4);
  if (threadIdx.x == 0) {
    nrm1[0] = t;
    const bool enough = t >= 0.5 * c1[i];
    skip[0] = enough ? 1.0 : 0.0;
    if (!enough && counter) counter[0] += 1.0;
  }
}

// Column j of H and the next basis vector:  h = c1 (+ c2 if the second pass ran),  beta = ||w||,  v_{j+1} = w / beta.
// An (exactly or numerically) invariant subspace -- beta <= 1e-13 ||A v_j|| -- is recorded in brk like a Lanczos
// breakdown; the remaining launches of the run are no-ops and the host uses the leading block.
__global__ __launch_bounds__(256) void k_arnoldi_finish(const double* __restrict__ c1, const double* __restrict__ c2,
                                                        const double* __restrict__ skip,
                                                        const double* __restrict__ nrm1,
                                                        const double* __restrict__ nrm2, int j,
                                                        double* __restrict__ hcol, const double* __restrict__ w1,
                                                        const double* __restrict__ w2, double* __restrict__ v_out,
                                                        int64_t n, double* __restrict__ brk) {
  if (broken(brk)) return;
  const bool second = skip[0] == 0.0;
  const double beta = sqrt(second ? nrm2[0] : nrm1[0]);
  const double scale = sqrt(c1[j + 1]);   // ||A v_j - shift v_j||
  const bool dead = !(beta > DSEA_BREAK_TOL * scale);
  if (blockIdx.x == 0) {
    for (int t = threadIdx.x; t <= j; t += 256) hcol[t] = second ? c1[t] + c2[t] : c1[t];
    if (threadIdx.x == 0) {
      hcol[j + 1] = beta;
      if (dead && brk) brk[0] = (double)(j + 1);
    }
  }
  if (dead) return;
  const double* __restrict__ w = second ? w2 : w1;
  const int64_t stride = (int64_t)gridDim.x * 512;
  for (int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2; row < n; row += stride) {
    double2 v = ld2<true>(w, row, n);
    v.x = v.x / beta;
    v.y = v.y / beta;
    st2<true>(v_out, row, n, v);
  }
}

// The same with the DGKS test inside (optimistic mode of dsea_arnoldi_extend: no second pass is enqueued).  Every block sums
// the correction pass's `count` partials of ||w1||^2 in the order of k_dgks_decide (same value in every block, bit for
// bit what the two-kernel sequence computes) and tests it; a step that NEEDS the second pass is not finished: block 0
// records brk[0] = -(j + 1) -- every later launch of the run is then a no-op (broken()) -- and dsea_arnoldi_status hands
// the step back to the caller, who repeats it with the second pass enqueued.
__global__ __launch_bounds__(256) void k_arnoldi_finish_opt(const double* __restrict__ c1, const double* __restrict__ P,
                                                            int count, int j, double* __restrict__ hcol,
                                                            const double* __restrict__ w1, double* __restrict__ v_out,
                                                            int64_t n, double* __restrict__ brk,
                                                            double* __restrict__ counter) {
  __shared__ double sm4[4];
  __shared__ double s_nrm1;
  if (broken(brk)) return;
  double acc = 0.0;
  for (int b = threadIdx.x; b < count; b += 256) acc += P[b];
  const double t0 = block_sum(acc, sm4);  // (the total is thread 0's)
  if (threadIdx.x == 0) s_nrm1 = t0;
  __syncthreads();
  const double nrm1 = s_nrm1;
  const double ww = c1[j + 1];            // ||A v_j - shift v_j||^2
  if (!(nrm1 >= 0.5 * ww)) {              // DGKS: the first pass lost too much -- this step needs the second pass
    if (blockIdx.x == 0 && threadIdx.x == 0) brk[0] = -(double)(j + 1);
    // (not counted here: the caller repeats the step in the default mode, whose k_dgks_decide counts the second pass once)
    return;
  }
  const double beta = sqrt(nrm1);
  const bool dead = !(beta > DSEA_BREAK_TOL * sqrt(ww));
  if (blockIdx.x == 0) {
    for (int t = threadIdx.x; t <= j; t += 256) hcol[t] = c1[t];
    if (threadIdx.x == 0) {
      hcol[j + 1] = beta;
      if (dead && brk) brk[0] = (double)(j + 1);
    }
  }
  if (dead) return;
  const int64_t stride = (int64_t)gridDim.x * 512;
  for (int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2; row < n; row += stride) {
    double2 v = ld2<true>(w1, row, n);
    v.x = v.x / beta;
    v.y = v.y / beta;
    st2<true>(v_out, row, n, v);
  }
}

// ------------------------------------------------------------------------------------------
// GMRES cycle scalars (restart length m <= 64): one thread each, all on the stream
// gw = device work: H (m+1) x m column-major | cs[m] | sn[m] | g[m+1] | y[m]
// state: [0] residual estimate  [1] converged (0/1)  [2] columns processed in this cycle  [3] ||r0||
//        [4] cycle finished early (converged, or the Krylov space is exhausted)
// brk  : the skip / breakdown record handed to the Arnoldi step kernels of the cycle (non-zero = steps are no-ops)
// ------------------------------------------------------------------------------------------
__global__ void k_gmres_begin(const double* __restrict__ nrm2, double target, double* __restrict__ g, int m,
                              double* __restrict__ state, double* __restrict__ brk) {
  const double beta0 = sqrt(nrm2[0]);
  for (int t = 0; t <= m; ++t) g[t] = 0.0;
  g[0] = beta0;
  state[0] = beta0;
  state[3] = beta0;
  state[2] = 0.0;
  const bool conv = beta0 <= target;
  state[1] = conv ? 1.0 : 0.0;
  state[4] = conv ? 1.0 : 0.0;
  state[5] = 0.0;              // (optimistic mode: a step of this cycle needed the second Gram-Schmidt pass)
  brk[0] = conv ? 1.0 : 0.0;   // converged already: every step kernel of the cycle returns at once
  brk[1] = 0.0;
}

// Givens update for column j (already written by the Arnoldi step unless the cycle had finished)
__global__ void k_gmres_givens(double* __restrict__ H, int ldh, int j, double* __restrict__ cs,
                               double* __restrict__ sn, double* __restrict__ g, double target,
                               double* __restrict__ state, double* __restrict__ brk) {
  if (state[4] != 0.0) return;
  if (brk[0] < 0.0) {
    // optimistic mode (dsea_ws_set_arnoldi_optimistic): step j failed the DGKS test and was not finished -- column j does
    // not exist.  The cycle ends here with the j columns it has (a restart, always valid); state[5] tells the caller to
    // run the next cycle with the second pass enqueued.
    state[4] = 1.0;
    state[5] = 1.0;
    return;
  }
  double* h = H + (int64_t)j * ldh;
  for (int t = 0; t < j; ++t) {
    const double a = h[t], b = h[t + 1];
    h[t] = cs[t] * a + sn[t] * b;
    h[t + 1] = -sn[t] * a + cs[t] * b;
  }
  const double rho = hypot(h[j], h[j + 1]);
  const double c = rho == 0.0 ? 1.0 : h[j] / rho, s = rho == 0.0 ? 0.0 : h[j + 1] / rho;
  cs[j] = c;
  sn[j] = s;
  h[j] = rho;
  h[j + 1] = 0.0;
  g[j + 1] = -s * g[j];
  g[j] = c * g[j];
  const double res = fabs(g[j + 1]);
  state[0] = res;
  state[2] = (double)(j + 1);
  if (res <= target) {
    state[1] = 1.0;
    state[4] = 1.0;
    brk[0] = 1.0;
  } else if (brk[0] != 0.0) {
    state[4] = 1.0;   // this step's finish kernel found the Krylov space exhausted: column j was the last one
  }
}

// back-substitution R y = g over the `steps` columns done; y[t >= steps] = 0
__global__ void k_gmres_solve(const double* __restrict__ H, int ldh, int m, const double* __restrict__ g,
                              const double* __restrict__ state, double* __restrict__ y) {
  const int steps = (int)state[2];
  for (int t = 0; t < m; ++t) y[t] = 0.0;
  for (int i = steps - 1; i >= 0; --i) {
    double acc = g[i];
    for (int t = i + 1; t < steps; ++t) acc -= H[(int64_t)t * ldh + i] * y[t];
    y[i] = acc / H[(int64_t)i * ldh + i];
  }
}

// r = b - u ; partial r.r
__global__ __launch_bounds__(256) void k_residual(const double* __restrict__ b, const double* __restrict__ u,
                                                  double* __restrict__ r, int64_t n, double* __restrict__ P) {
  __shared__ double sm4[4];
  double acc = 0.0;
  const int64_t stride = (int64_t)gridDim.x * 512;
  for (int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2; row < n; row += stride) {
    double2 bv = ld2<true>(b, row, n);
    if (u) {
      const double2 uv = ld2<true>(u, row, n);
      bv.x -= uv.x;
      bv.y -= uv.y;
    }
    st2<true>(r, row, n, bv);
    acc = fma(bv.x, bv.x, acc);
    acc = fma(bv.y, bv.y, acc);
  }
  double t = block_sum(acc, sm4);
  if (threadIdx.x == 0) P[blockIdx.x] = t;
}

static inline int kr_blocks(int64_t n) {
  int64_t nb = (n + 2047) / 2048;
  if (nb < 1) nb = 1;
  if (nb > DSEA_MAX_EW_BLOCKS) nb = DSEA_MAX_EW_BLOCKS;
  return (int)nb;
}

// One Arnoldi step j -> j+1 on (A - shift I):  w = A v_j - shift v_j, orthogonalised against V[0..j] (CGS + DGKS
// second pass), column j of H, v_{j+1}.  brk: break / skip record of the run (2 doubles), skip: DGKS flag (1 double).
int arnoldi_step(const OpDesc& op, Workspace& w, const double* shift_or_zero, double* V, int64_t ldv, int j,
                 double* hcol, double* brk, double* skip, double* nrm1, double* nrm2, hipStream_t st, bool optimistic) {
  double* u = w.vec[0];
  const double* vj = V + (int64_t)j * ldv;
  int nb = launch_spmv(op, vj, u, nullptr, brk, nullptr, st);
  if (nb < 0) return -1;
  arnoldi_orth(w, op.n, u, shift_or_zero, V, ldv, j, hcol, brk, skip, nrm1, nrm2, st, optimistic);
  return 0;
}

// the step without its mat-vec: u = A v_j is given (generic-callable mode: the mat-vec is the caller's code)
void arnoldi_orth(Workspace& w, int64_t n, const double* u, const double* shift_or_zero, double* V, int64_t ldv, int j,
                  double* hcol, double* brk, double* skip, double* nrm1, double* nrm2, hipStream_t st, bool optimistic) {
  double* w1 = w.vec[1];
  double* w2 = w.vec[2];
  double* c1 = w.coef;
  double* c2 = w.coef2;
  const TileGeom g = w.geom(n);
  const int i = j + 1;
  // pass 1: w1 = u - shift v_j ; c1 = V^T w1 ; c1[i] = ||w1||^2 ; w1 -= V c1 ; nrm1 = ||w1||^2
  launch_rdots(g, V, ldv, n, i, u, shift_or_zero, nullptr, w1, w.partials, c1, st, nullptr, nullptr, 0, nullptr, true, brk);
  launch_axpy_norm(g, V, ldv, n, i, c1, w1, w.partials, nullptr, st, nullptr, brk);
  if (optimistic) {
    // six launches per step instead of eleven: the DGKS test rides in the finish kernel, the second pass (needed on 0 of 200
    // steps of the D = 512 transfer matrix) is not enqueued; a step that needs it hands itself back (k_arnoldi_finish_opt)
    hipLaunchKernelGGL(k_arnoldi_finish_opt, dim3(kr_blocks(n)), dim3(256), 0, st, (const double*)c1,
                       (const double*)w.partials, g.nw, j, hcol, (const double*)w1, V + (int64_t)(j + 1) * ldv, n, brk,
                       w.scal + 31);
    return;
  }
  hipLaunchKernelGGL(k_dgks_decide, dim3(1), dim3(256), 0, st, (const double*)c1, i, (const double*)w.partials, g.nw,
                     nrm1, skip, (const double*)brk, w.scal + 31);
  // pass 2 (skipped on the device unless the DGKS test failed): w2 = w1 - V (V^T w1)
  launch_rdots(g, V, ldv, n, i, w1, w.zero, nullptr, w2, w.partials, c2, st, nullptr, nullptr, 0, nullptr, false, skip);
  launch_axpy_norm(g, V, ldv, n, i, c2, w2, w.partials, nullptr, st, nullptr, skip);
  launch_finalize_slot(w.partials, g.nw, nrm2, skip, st);
  hipLaunchKernelGGL(k_arnoldi_finish, dim3(kr_blocks(n)), dim3(256), 0, st, (const double*)c1, (const double*)c2,
                     (const double*)skip, (const double*)nrm1, (const double*)nrm2, j, hcol, (const double*)w1,
                     (const double*)w2, V + (int64_t)(j + 1) * ldv, n, brk);
}

void launch_residual(const double* b, const double* u, double* r, int64_t n, double* P, double* nrm2_out,
                     hipStream_t st) {
  const int nb = kr_blocks(n);
  hipLaunchKernelGGL(k_residual, dim3(nb), dim3(256), 0, st, b, u, r, n, P);
  launch_finalize1(P, nb, nrm2_out, st);
}

void launch_gmres_begin(const double* nrm2, double target, double* g, int m, double* state, double* brk,
                        hipStream_t st) {
  hipLaunchKernelGGL(k_gmres_begin, dim3(1), dim3(1), 0, st, nrm2, target, g, m, state, brk);
}
void launch_gmres_givens(double* H, int ldh, int j, double* cs, double* sn, double* g, double target, double* state,
                         double* brk, hipStream_t st) {
  hipLaunchKernelGGL(k_gmres_givens, dim3(1), dim3(1), 0, st, H, ldh, j, cs, sn, g, target, state, brk);
}
void launch_gmres_solve(const double* H, int ldh, int m, const double* g, const double* state, double* y,
                        hipStream_t st) {
  hipLaunchKernelGGL(k_gmres_solve, dim3(1), dim3(1), 0, st, H, ldh, m, g, state, y);
}

}  // namespace dsea
