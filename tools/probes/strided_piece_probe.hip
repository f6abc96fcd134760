// Stand-alone probe (not part of the library): how fast does the chip move a 2^25-row fp64 vector when every workgroup
// touches it as NH pieces of PD doubles at a stride of 2^SB rows -- the access shape of a "high tile bits" pass of a two-pass
// TFIM mat-vec?  Each workgroup takes one column piece c and all NH high indices h:
//     dst[h * 2^SB + c * PD + e] = 2 * src[same]          (read once, write once: 2 * 268 MB per launch)
//   hipcc --offload-arch=gfx950 -O3 -o strided_piece_probe strided_piece_probe.hip && ./strided_piece_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int PD>
__global__ void __launch_bounds__(1024) k_pieces(const double* __restrict__ src, double* __restrict__ dst, int sb, int nh) {
  constexpr int TPP = PD / 2;                              // threads per piece (one double2 each)
  const int64_t c = blockIdx.x;
  const int e = threadIdx.x % TPP, h0 = threadIdx.x / TPP, hstep = 1024 / TPP;
  const int64_t base = c * PD + 2 * e;
#pragma unroll 4
  for (int h = h0; h < nh; h += hstep) {
    const int64_t at = ((int64_t)h << sb) + base;
    double2 v = *reinterpret_cast<const double2*>(src + at);
    v.x *= 2.0; v.y *= 2.0;
    *reinterpret_cast<double2*>(dst + at) = v;
  }
}
__global__ void __launch_bounds__(1024) k_copy(const double* __restrict__ src, double* __restrict__ dst, int64_t n) {
  for (int64_t i = ((int64_t)blockIdx.x * 1024 + threadIdx.x) * 2; i < n; i += (int64_t)gridDim.x * 2048) {
    double2 v = *reinterpret_cast<const double2*>(src + i);
    v.x *= 2.0; v.y *= 2.0;
    *reinterpret_cast<double2*>(dst + i) = v;
  }
}
template <class F>
static float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); f();
  hipEventRecord(e0, 0);
  for (int r = 0; r < 10; ++r) f();
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
  return ms / 10.f;
}
int main() {
  const int L = 25; const int64_t n = (int64_t)1 << L;
  double *a, *b; hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMemset(a, 0, n * 8); hipMemset(b, 0, n * 8);
  const double gb = 2.0 * n * 8 / 1e9;
  float ms = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(4096), dim3(1024), 0, 0, a, b, n); });
  printf("contiguous copy                                   %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
  for (int sb = 14; sb <= 16; ++sb) {
    const int nh = 1 << (L - sb);
#define RUN(PD) { const int64_t wgs = ((int64_t)1 << sb) / PD; \
    ms = timeit([&] { hipLaunchKernelGGL(k_pieces<PD>, dim3((unsigned)wgs), dim3(1024), 0, 0, a, b, sb, nh); }); \
    printf("stride 2^%d rows, %5d pieces of %3d B per workgroup, %5lld workgroups: %.3f ms  %.0f GB/s\n", sb, nh, PD * 8, (long long)wgs, ms, gb / ms * 1e3); }
    RUN(8) RUN(16) RUN(32) RUN(64) RUN(128)
  }
  return 0;
}
