// dsea_device.h -- device-side helpers shared by the kernel translation units (dsea_kernels.hip, dsea_krylov.hip).
#ifndef DSEA_DEVICE_H
#define DSEA_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dsea {

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
// Four wave-wide sums at once without the LDS crossbar: on gfx950 ``__shfl_xor`` is a ds_bpermute (a round trip
// through the LDS pipeline, ~100+ cycles, seven of them in a dependent chain for the transposed butterfly the dots
// pass used) during which a wave that is alone on its SIMD has no loads in flight.  v_permlane32_swap / v_permlane16_swap
// (new in gfx950) exchange half-waves / alternate 16-lane rows of two registers in one VALU instruction, which is
// exactly the "transposed" step: after swap32(a0, a2) the sum a0 + a2 holds the 32-lane partial sums of a0 in its lower
// half and those of a2 in its upper half (likewise a1, a3); after swap16 of the two results, row v (lanes 16v..16v+15)
// of their sum holds 16 partial sums of a_v; four DPP row shifts finish inside the rows.
// Returns the total of a_v in lane 16 v + 15 (other lanes hold partial sums).  Fixed order: deterministic.
typedef unsigned dsea_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void permlane32_swap_f64(double& d, double& s) {
  const dsea_v2u lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(d), (unsigned)__double2loint(s), false, false);
  const dsea_v2u hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(d), (unsigned)__double2hiint(s), false, false);
  d = __hiloint2double((int)hi.x, (int)lo.x);
  s = __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ void permlane16_swap_f64(double& d, double& s) {
  const dsea_v2u lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(d), (unsigned)__double2loint(s), false, false);
  const dsea_v2u hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(d), (unsigned)__double2hiint(s), false, false);
  d = __hiloint2double((int)hi.x, (int)lo.x);
  s = __hiloint2double((int)hi.y, (int)lo.y);
}
template <int CTRL>   // DPP row_shr:n = 0x110 + n ; lanes without a source read 0
__device__ __forceinline__ double dpp_row_shr_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum4_rows(double a0, double a1, double a2, double a3) {
  permlane32_swap_f64(a0, a2);
  permlane32_swap_f64(a1, a3);
  double A = a0 + a2, B = a1 + a3;
  permlane16_swap_f64(A, B);
  double v = A + B;
  v += dpp_row_shr_f64<0x118>(v);
  v += dpp_row_shr_f64<0x114>(v);
  v += dpp_row_shr_f64<0x112>(v);
  v += dpp_row_shr_f64<0x111>(v);
  return v;
}

// One wave-wide sum, the total in EVERY lane, without the LDS crossbar (six dependent ds_bpermute round trips before):
// half-waves and alternate rows are folded with the gfx950 permlane swaps (both registers hold the same value, so
// all lanes keep a valid partial sum), the 16 lanes of a row with four DPP row rotations.  Fixed order: deterministic.
template <int CTRL>   // DPP row_ror:n = 0x120 + n
__device__ __forceinline__ double dpp_row_ror_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
  double a = v, b = v;
  permlane32_swap_f64(a, b);
  v = a + b;                       // lane i: x[i mod 32] + x[i mod 32 + 32]
  a = v;
  b = v;
  permlane16_swap_f64(a, b);
  v = a + b;                       // lane i: the four rows' values of column i mod 16
  v += dpp_row_ror_f64<0x128>(v);
  v += dpp_row_ror_f64<0x124>(v);
  v += dpp_row_ror_f64<0x122>(v);
  v += dpp_row_ror_f64<0x121>(v);
  return v;                        // every lane holds the same, order-fixed total
}

// wave-wide maximum in every lane, the same way
__device__ __forceinline__ double wave_max(double v) {
  double a = v, b = v;
  permlane32_swap_f64(a, b);
  v = fmax(a, b);
  a = v;
  b = v;
  permlane16_swap_f64(a, b);
  v = fmax(a, b);
  v = fmax(v, dpp_row_ror_f64<0x128>(v));
  v = fmax(v, dpp_row_ror_f64<0x124>(v));
  v = fmax(v, dpp_row_ror_f64<0x122>(v));
  v = fmax(v, dpp_row_ror_f64<0x121>(v));
  return v;
}

// block of 256 threads = 4 waves; returns the total in thread 0 (fixed order w0+w1+w2+w3)
__device__ __forceinline__ double block_sum(double v, double* sm4) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sm4[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) t = ((sm4[0] + sm4[1]) + sm4[2]) + sm4[3];
  return t;
}

template <bool GUARD>
__device__ __forceinline__ double2 ld2(const double* __restrict__ p, int64_t row, int64_t n) {
  if (!GUARD || row + 1 < n) return *reinterpret_cast<const double2*>(p + row);
  double2 v = make_double2(0.0, 0.0);
  if (row < n) v.x = p[row];
  return v;
}
// streaming load of basis data that is read once per pass: non-temporal hint (does not displace r / partials
// in L2)
template <bool GUARD>
__device__ __forceinline__ double2 ld2_stream(const double* __restrict__ p, int64_t row, int64_t n) {
  // measured on MI355X (tools/kbench.py, n = 2^20, i = 199): dots pass 290 us -> 255 us with the nt hint
#ifdef DSEA_NO_NT   /* A/B build (make libdsea_nont.so EXTRA=-DDSEA_NO_NT): default cache policy on the basis stream */
  return ld2<GUARD>(p, row, n);
#endif
  if (!GUARD || row + 1 < n) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    v2d t = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p + row));
    return make_double2(t.x, t.y);
  }
  return ld2<GUARD>(p, row, n);
}
__device__ __forceinline__ uint4 ld_u4_stream(const uint16_t* __restrict__ p) {
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
  v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p));
  return make_uint4(t.x, t.y, t.z, t.w);
}

template <bool GUARD>
__device__ __forceinline__ void st2(double* __restrict__ p, int64_t row, int64_t n, double2 v) {
  if (!GUARD || row + 1 < n) {
    *reinterpret_cast<double2*>(p + row) = v;
  } else if (row < n) {
    p[row] = v.x;
  }
}

// bf16 shadow of the basis (storage only, see k_axpy_norm_lp): fp64 -> bf16 round-to-nearest-even
__device__ __forceinline__ uint16_t f64_to_bf16(double v) {
  const uint32_t u = __float_as_uint((float)v);
  return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ double bf16lo_to_f64(uint32_t packed) { return (double)__uint_as_float(packed << 16); }
__device__ __forceinline__ double bf16hi_to_f64(uint32_t packed) {
  return (double)__uint_as_float(packed & 0xFFFF0000u);
}
__device__ __forceinline__ void st_bf16x2(uint16_t* __restrict__ p, int64_t row, int64_t n, double2 v) {
  if (row + 1 < n) {
    *reinterpret_cast<uint32_t*>(p + row) = (uint32_t)f64_to_bf16(v.x) | ((uint32_t)f64_to_bf16(v.y) << 16);
  } else if (row < n) {
    p[row] = f64_to_bf16(v.x);
  }
}

// Breakdown record of a native Lanczos run: brk[0] = step at which beta ~ 0 was found (0 = none),
// brk[1] = running max of |alpha_j|, |beta_j| (the scale beta is compared with).  Every kernel of the loop
// starts with broken(brk): once the record is set the remaining launches of the run are no-ops.
#define DSEA_BREAK_TOL 1e-13
__device__ __forceinline__ bool broken(const double* __restrict__ brk) { return brk && brk[0] != 0.0; }

// Consumers that fold the second reduction stage into their prologue: every wave / block sums the same
// partials in the same order, so all of them obtain the bit-identical scalar without a separate launch.
__device__ __forceinline__ double sum_partials_wave(const double* __restrict__ P, int count, int lane) {
  double a0 = 0.0, a1 = 0.0;
  int b = lane;
  for (; b + 64 < count; b += 128) {
    a0 += P[b];
    a1 += P[b + 64];
  }
  if (b < count) a0 += P[b];
  return wave_sum(a0 + a1);
}
__device__ __forceinline__ double sum_partials_block(const double* __restrict__ P, int count, double* sm5) {
  double a0 = 0.0, a1 = 0.0;
  int b = threadIdx.x;
  for (; b + 256 < count; b += 512) {
    a0 += P[b];
    a1 += P[b + 256];
  }
  if (b < count) a0 += P[b];
  double v = wave_sum(a0 + a1);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sm5[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) sm5[4] = ((sm5[0] + sm5[1]) + sm5[2]) + sm5[3];
  __syncthreads();
  return sm5[4];
}

// ------------------------------------------------------------------------------------------
// Data-tagged granules: the cross-workgroup exchange of the persistent single-launch solvers (k_cg_persist_stencil*,
// k_cg_persist_tfim*, k_lanczos_persist).  A double travels as two 8-byte words, each carrying 32 bits of data and the
// 32-bit epoch of the exchange; a reader accepts the value when both words carry the epoch it waits for.  Relaxed
// agent-scope stores and polls, no fences, no separate flags (cdna_hip_programming.md Guideline 16, form R2); the
// buffers are zeroed by the launcher before every launch (epoch 0 = "nothing yet").
// ------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) unsigned long long gran_u64;
#define DSEA_GRANULE_TIMEOUT_TICKS 300000000ll /* 3 s of the 100 MHz wall clock: a lost peer must not hang the GPU */
__device__ __forceinline__ void granule_put(gran_u64* g, unsigned epoch, double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned long long tag = (unsigned long long)epoch << 32;
  __hip_atomic_store(g, tag | (b & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(g + 1, tag | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool granule_try_get(gran_u64* g, unsigned epoch, double& v) {
  const unsigned long long lo = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long hi = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v = __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffull)));
  return (unsigned)(lo >> 32) == epoch && (unsigned)(hi >> 32) == epoch;
}
// the epoch a granule currently carries (the smaller of its two words' tags)
__device__ __forceinline__ unsigned granule_epoch(gran_u64* g) {
  const unsigned long long lo = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long hi = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned a = (unsigned)(lo >> 32), b = (unsigned)(hi >> 32);
  return a < b ? a : b;
}

}  // namespace dsea
#endif
