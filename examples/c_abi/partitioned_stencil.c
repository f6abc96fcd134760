/* A plain-C consumer of the ROW-PARTITIONED entry points of include/dsea.h (dsea_comm_*, dsea_pop_*): what a host program
 * that owns its own transport (MPI, sockets, ...) binds -- no Python, no torch, no RCCL.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_abi/partitioned_stencil.c \
 *       -Ldominantsparseeigenad_amd/csrc -ldsea -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/dominantsparseeigenad_amd/csrc \
 *       -o partitioned_stencil && ./partitioned_stencil [N]
 *
 * Two communicator forms on ONE rank (a one-GPU box has nothing else to offer without a transport):
 *   (a) dsea_comm_create_callbacks with the caller's own collectives (with one rank the library has nothing to reduce or to
 *       exchange and does not call them; with more ranks every all-reduce / halo exchange of the solvers comes back here);
 *   (b) dsea_comm_unique_id + dsea_comm_init_rank: communicators the library creates over the RCCL of the process
 *       (skipped with a note where no RCCL can be loaded).
 * On each: dsea_pop_create_stencil3 for A = tridiag(-1, 2, -1) (coef = -1, V = 0, Dirichlet ends), k = N steps of
 * dsea_pop_lanczos_run -- the ends of T's spectrum against 2 - 2 cos(j pi / (N + 1)) -- and dsea_pop_cg_run on
 * (A + 1) x = b with the residual formed by dsea_pop_matvec.                                                       */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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

static int g_allreduces = 0;
/* the caller's transport: with one rank a sum over ranks is the value itself (a real program would call MPI_Allreduce on a
 * staged copy, or its own device-aware collective, on `stream`) */
static int my_allreduce(void *user, double *buf, int64_t count, void *stream) {
  (void)user; (void)buf; (void)count; (void)stream;
  ++g_allreduces;
  return 0;
}
static int my_alltoall(void *user, const double *send, double *recv, int64_t chunk, void *stream) {
  (void)user;
  return hipMemcpyAsync(recv, send, (size_t)chunk * sizeof(double), hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? 0 : 1;
}
static int my_sendrecv(void *user, const double *send, double *recv, int64_t count, int peer, void *stream) {
  (void)user; (void)send; (void)recv; (void)count; (void)peer; (void)stream;
  return 1;                                                    /* one rank has no peer: never called */
}

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

/* Lanczos + CG on one communicator; returns 0 on PASS */
static int run_on(dsea_comm_t comm, const char *what, int64_t n) {
  const int k = (int)n;
  const int64_t ldq = (n + 31) / 32 * 32;
  const double pi = 3.14159265358979323846;
  double *V = NULL, *halo = NULL, *q0 = NULL, *Q = NULL, *alphas = NULL, *betas = NULL, *b = NULL, *x = NULL, *y = NULL, *state = NULL,
         *shift = NULL, *dot = NULL;
  void *wsbuf = NULL;
  size_t wsbytes = 0;
  HIP_OK(hipMalloc((void **)&V, (size_t)n * sizeof(double)));
  HIP_OK(hipMemset(V, 0, (size_t)n * sizeof(double)));
  HIP_OK(hipMalloc((void **)&halo, 2 * sizeof(double)));
  HIP_OK(hipMemset(halo, 0, 2 * sizeof(double)));
  HIP_OK(hipMalloc((void **)&q0, (size_t)n * sizeof(double)));
  HIP_OK(hipMalloc((void **)&b, (size_t)n * sizeof(double)));
  HIP_OK(hipMalloc((void **)&x, (size_t)n * sizeof(double)));
  HIP_OK(hipMalloc((void **)&y, (size_t)n * sizeof(double)));
  HIP_OK(hipMalloc((void **)&Q, (size_t)k * (size_t)ldq * sizeof(double)));
  HIP_OK(hipMalloc((void **)&alphas, (size_t)k * sizeof(double)));
  HIP_OK(hipMalloc((void **)&betas, (size_t)k * sizeof(double)));
  HIP_OK(hipMalloc((void **)&state, DSEA_CG_STATE_LEN * sizeof(double)));
  HIP_OK(hipMalloc((void **)&shift, sizeof(double)));
  HIP_OK(hipMalloc((void **)&dot, sizeof(double)));
  double *h = (double *)malloc((size_t)n * sizeof(double));
  unsigned long long s = 88172645463325252ull;
  for (int64_t i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    h[i] = (double)(s >> 11) / 9007199254740992.0 - 0.5;
  }
  HIP_OK(hipMemcpy(q0, h, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(b, h, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
  HIP_OK(hipMemset(x, 0, (size_t)n * sizeof(double)));
  const double minus_one = -1.0;                                /* (A - shift) with shift = -1: A + 1, positive definite */
  HIP_OK(hipMemcpy(shift, &minus_one, sizeof(double), hipMemcpyHostToDevice));

  dsea_pop_t pop = NULL;
  dsea_ws_t ws = NULL;
  DSEA_OK_OR_DIE(dsea_pop_create_stencil3(n, -1.0, V, halo, comm, &pop));
  DSEA_OK_OR_DIE(dsea_ws_bytes(n, k, &wsbytes));
  HIP_OK(hipMalloc(&wsbuf, wsbytes));
  DSEA_OK_OR_DIE(dsea_ws_create(wsbuf, wsbytes, n, k, &ws));

  /* forward: reference Lanczos.py:49-77, the collectives issued by the library */
  DSEA_OK_OR_DIE(dsea_pop_lanczos_run(pop, ws, k, q0, Q, ldq, alphas, betas, NULL));
  int step = 0;
  DSEA_OK_OR_DIE(dsea_pop_lanczos_status(pop, ws, &step, NULL));                        /* synchronises */
  double *a = (double *)malloc((size_t)k * sizeof(double)), *bb = (double *)malloc((size_t)k * sizeof(double));
  HIP_OK(hipMemcpy(a, alphas, (size_t)k * sizeof(double), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(bb, betas, (size_t)(k - 1) * sizeof(double), hipMemcpyDeviceToHost));
  const double lo = kth_eigenvalue(a, bb, k, 0, -1.0, 5.0), hi = kth_eigenvalue(a, bb, k, k - 1, -1.0, 5.0);
  const double lo_ref = 2.0 - 2.0 * cos(pi / (double)(n + 1)), hi_ref = 2.0 - 2.0 * cos((double)n * pi / (double)(n + 1));
  const double err = fmax(fabs(lo - lo_ref), fabs(hi - hi_ref));

  /* adjoint-type solve: reference CG.py:24-41 on (A + 1) x = b, then the residual through dsea_pop_matvec */
  int64_t iters = 0;
  double resnorm = 0.0;
  const int rc = dsea_pop_cg_run(pop, ws, shift, b, x, state, 1e-10, n, 8, &iters, &resnorm, NULL);
  if (rc != DSEA_OK) {
    fprintf(stderr, "dsea_pop_cg_run: %s\n", dsea_error_string(rc));
    return 3;
  }
  DSEA_OK_OR_DIE(dsea_pop_matvec(pop, ws, x, y, shift, dot, NULL, NULL));               /* y = (A + 1) x, dot = x.y (global) */
  HIP_OK(hipDeviceSynchronize());
  double *hy = (double *)malloc((size_t)n * sizeof(double));
  HIP_OK(hipMemcpy(hy, y, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  double res2 = 0.0;
  for (int64_t i = 0; i < n; ++i) res2 += (hy[i] - h[i]) * (hy[i] - h[i]);
  const double res = sqrt(res2);
  printf("%s: spectrum ends off by %.2e; CG %lld iterations, reported residual %.2e, true residual %.2e\n", what, err,
         (long long)iters, resnorm, res);

  DSEA_OK_OR_DIE(dsea_ws_destroy(ws));
  DSEA_OK_OR_DIE(dsea_pop_destroy(pop));
  hipFree(wsbuf); hipFree(V); hipFree(halo); hipFree(q0); hipFree(Q); hipFree(alphas); hipFree(betas); hipFree(b); hipFree(x);
  hipFree(y); hipFree(state); hipFree(shift); hipFree(dot);
  free(h); free(a); free(bb); free(hy);
  return (err < 1e-12 && res < 1e-8) ? 0 : 1;
}

int main(int argc, char **argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 300;
  printf("libdsea %d: row-partitioned entry points, A = tridiag(-1, 2, -1), N = %lld on one rank\n", dsea_version(), (long long)n);
  HIP_OK(hipSetDevice(0));
  int fail = 0;

  dsea_comm_t comm = NULL;
  DSEA_OK_OR_DIE(dsea_comm_create_callbacks(0, 1, my_allreduce, my_alltoall, my_sendrecv, NULL, &comm));
  fail |= run_on(comm, "caller-supplied collectives", n);
  DSEA_OK_OR_DIE(dsea_comm_destroy(comm));
  /* (with ONE rank the library does not call the all-reduce back: a sum over one rank is the value itself) */
  printf("the library called back for %d all-reduces\n", g_allreduces);

  unsigned char id[DSEA_COMM_ID_BYTES];
  const int rc = dsea_comm_unique_id(id);
  if (rc == DSEA_ERR_UNSUPPORTED) {
    puts("library-owned RCCL communicators: no RCCL can be loaded in this process -- skipped");
  } else {
    if (rc != DSEA_OK) {
      fprintf(stderr, "dsea_comm_unique_id: %s\n", dsea_error_string(rc));
      return 3;
    }
    DSEA_OK_OR_DIE(dsea_comm_init_rank(id, NULL, 0, 1, &comm));
    fail |= run_on(comm, "library-owned RCCL communicator", n);
    DSEA_OK_OR_DIE(dsea_comm_destroy(comm));
  }
  puts(fail ? "FAIL" : "PASS");
  return fail;
}
