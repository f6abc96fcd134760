/* A plain-C consumer of include/dsea.h: the calls a host program (or the cgo / JNI / ctypes stub of INTEGRATION.md) makes for
 * reference Lanczos.py:49-77 on a native operand -- no Python, no torch, only the HIP runtime's C API for the device buffers.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_abi/lanczos_stencil.c \
 *       -Ldominantsparseeigenad_amd/csrc -ldsea -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/dominantsparseeigenad_amd/csrc \
 *       -o lanczos_stencil && ./lanczos_stencil [N]
 *
 * Operand: A = tridiag(-1, 2, -1) of order N (dsea_op_create_stencil3 with coef = -1, V = 0: the kinetic part of
 * schrodinger1D.py:18-27), whose eigenvalues are 2 - 2 cos(j pi / (N + 1)).  k = N Lanczos steps with the library's full
 * re-orthogonalisation give a tridiagonal T with exactly that spectrum; the program checks the two ends of it (bisection on
 * the Sturm sequence of T, on the host) and prints PASS / FAIL.                                                           */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "dsea.h"

#define HIP_OK(call)                                                                 \
  do {                                                                               \
    hipError_t e_ = (call);                                                          \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                     \
      return 2;                                                                      \
    }                                                                                \
  } while (0)
#define DSEA_OK_OR_DIE(call)                                                         \
  do {                                                                               \
    int s_ = (call);                                                                 \
    if (s_ != DSEA_OK) {                                                             \
      fprintf(stderr, "%s: %s\n", #call, dsea_error_string(s_));                     \
      return 3;                                                                      \
    }                                                                                \
  } while (0)

/* number of eigenvalues of the symmetric tridiagonal (a, b) below x */
static int sturm_count(const double *a, const double *b, int k, double x) {
  int count = 0;
  double d = 1.0;
  for (int i = 0; i < k; ++i) {
    const double off = i == 0 ? 0.0 : b[i - 1] * b[i - 1];
    d = (a[i] - x) - (i == 0 ? 0.0 : off / d);
    if (d == 0.0) d = 1e-300;
    if (d < 0.0) ++count;
  }
  return count;
}
static double kth_eigenvalue(const double *a, const double *b, int k, int which, double lo, double hi) {
  for (int it = 0; it < 200; ++it) {
    const double mid = 0.5 * (lo + hi);
    if (sturm_count(a, b, k, mid) > which) hi = mid; else lo = mid;
  }
  return 0.5 * (lo + hi);
}

int main(int argc, char **argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 300;
  const int k = (int)n;
  const int64_t ldq = (n + 31) / 32 * 32;
  const double pi = 3.14159265358979323846;
  printf("libdsea %d: A = tridiag(-1, 2, -1), N = %lld, k = %d Lanczos steps through the C ABI\n", dsea_version(), (long long)n, k);

  double *V = NULL, *q0 = NULL, *Q = NULL, *alphas = NULL, *betas = NULL;
  void *wsbuf = NULL;
  size_t wsbytes = 0;
  HIP_OK(hipSetDevice(0));
  HIP_OK(hipMalloc((void **)&V, (size_t)n * sizeof(double)));
  HIP_OK(hipMemset(V, 0, (size_t)n * sizeof(double)));
  HIP_OK(hipMalloc((void **)&q0, (size_t)n * sizeof(double)));
  HIP_OK(hipMalloc((void **)&Q, (size_t)k * (size_t)ldq * sizeof(double)));
  HIP_OK(hipMalloc((void **)&alphas, (size_t)k * sizeof(double)));
  HIP_OK(hipMalloc((void **)&betas, (size_t)k * sizeof(double)));
  double *h = (double *)malloc((size_t)n * sizeof(double));
  unsigned long long s = 88172645463325252ull;                 /* xorshift start vector */
  for (int64_t i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    h[i] = (double)(s >> 11) / 9007199254740992.0 - 0.5;
  }
  HIP_OK(hipMemcpy(q0, h, (size_t)n * sizeof(double), hipMemcpyHostToDevice));

  dsea_op_t op = NULL;
  dsea_ws_t ws = NULL;
  DSEA_OK_OR_DIE(dsea_op_create_stencil3(n, -1.0, V, NULL, NULL, &op));
  DSEA_OK_OR_DIE(dsea_ws_bytes(n, k, &wsbytes));
  HIP_OK(hipMalloc(&wsbuf, wsbytes));
  DSEA_OK_OR_DIE(dsea_ws_create(wsbuf, wsbytes, n, k, &ws));
  DSEA_OK_OR_DIE(dsea_lanczos_run(op, ws, k, q0, Q, ldq, alphas, betas, NULL));      /* stream 0 */
  int brk = 0;
  const int st = dsea_lanczos_status(ws, &brk, NULL);                                 /* synchronises */
  if (st != DSEA_OK) {
    fprintf(stderr, "dsea_lanczos_status: %s (step %d)\n", dsea_error_string(st), brk);
    return 4;
  }
  double *a = (double *)malloc((size_t)k * sizeof(double)), *b = (double *)malloc((size_t)k * sizeof(double));
  HIP_OK(hipMemcpy(a, alphas, (size_t)k * sizeof(double), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(b, betas, (size_t)(k - 1) * sizeof(double), hipMemcpyDeviceToHost));

  const double lo = kth_eigenvalue(a, b, k, 0, -1.0, 5.0), hi = kth_eigenvalue(a, b, k, k - 1, -1.0, 5.0);
  const double lo_ref = 2.0 - 2.0 * cos(pi / (double)(n + 1)), hi_ref = 2.0 - 2.0 * cos((double)n * pi / (double)(n + 1));
  const double err = fmax(fabs(lo - lo_ref), fabs(hi - hi_ref));
  printf("lowest  %.15e  (exact %.15e)\nhighest %.15e  (exact %.15e)\nmax deviation %.2e\n", lo, lo_ref, hi, hi_ref, err);
  DSEA_OK_OR_DIE(dsea_ws_destroy(ws));
  DSEA_OK_OR_DIE(dsea_op_destroy(op));
  hipFree(wsbuf); hipFree(V); hipFree(q0); hipFree(Q); hipFree(alphas); hipFree(betas);
  free(h); free(a); free(b);
  puts(err < 1e-12 ? "PASS" : "FAIL");
  return err < 1e-12 ? 0 : 1;
}
