// Stand-alone probe (not part of the library): what does v_mfma_f64_16x16x4_f64 sustain on this chip?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_probe mfma_f64_probe.hip && ./mfma_f64_probe
// Pure register loop, NACC independent accumulators per wave, WPS waves per SIMD, one workgroup per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k_probe(double* out, int iters, double a0, double b0) {
  v4d acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int threads, int blocks, double* out) {
  const int iters = 4096 / NACC;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_probe<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 2.0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
  }
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfmas_per_wave = (double)iters * NACC, waves = (double)blocks * threads / 64.0;
  const double flops = mfmas_per_wave * waves * 2.0 * 16 * 16 * 4;
  printf("threads/WG %4d  WGs %4d  independent accumulators %d: %.1f us  -> %.1f TFLOP/s fp64, %.1f ns per MFMA per wave (%.0f cycles at 2.4 GHz)\n",
         threads, blocks, NACC, ms * 1e3, flops / (ms * 1e-3) / 1e12, ms * 1e6 / mfmas_per_wave, ms * 1e6 / mfmas_per_wave * 2.4);
}
int main() {
  double* out;
  hipMalloc(&out, 1 << 24);
  for (int threads : {256, 512, 1024}) {
    run<1>(threads, 256, out);
    run<2>(threads, 256, out);
    run<4>(threads, 256, out);
    run<8>(threads, 256, out);
  }
  run<4>(256, 512, out);
  run<4>(256, 1024, out);
  return 0;
}
