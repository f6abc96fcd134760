// dsea_internal.h -- shared between dsea_kernels.hip (device code + launchers) and dsea_capi.hip (C ABI).
#ifndef DSEA_INTERNAL_H
#define DSEA_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dsea.h"

#define DSEA_MAX_EW_BLOCKS 2048   /* grid cap of the grid-stride streaming kernels            */
#define DSEA_MAX_WAVE_TILES 8192  /* cap on wave tiles (= partial sums per basis vector)      */
#define DSEA_TFIM_TILE_LOG2 11    /* rows of x staged in LDS per block of the TFIM mat-vec    */
#define DSEA_MAX_TFIM_BLOCKS 4096 /* grid cap of the TFIM mat-vec (<= DSEA_MAX_WAVE_TILES partial slots)   */
#define DSEA_PERSIST_MAX_TILES 4096 /* canonical-tile regime of the CG kernels: n <= 2^21 rows               */
#define DSEA_MAX_KRYLOV 8000       /* cap on kmax: the dots pass keeps one LDS row of k partial sums per wave      */
#define DSEA_PERSIST_CG_MAX_TILES 1024 /* persistent single-launch CG: n <= 2^19 rows                       */
#define DSEA_SCALARS 64
#define DSEA_SCAL_BREAK 20    /* scal[20] = breakdown step, scal[21] = running scale (see broken())     */
#define DSEA_SCAL_PRO 40      /* scal[40] = re-orthogonalise this step, [41] = and the next, [42] = ||A|| estimate, [43] = steps re-orthogonalised, [44] = global ||r||^2 (row-partitioned run) */
/* default threshold of the partial re-orthogonalisation on the estimated |q_i . q_k|: the path's stated tolerance.  Simon's
 * classical sqrt(eps) = 1.5e-8 keeps the Ritz VALUES at full accuracy; the corrections that are dropped from T (Q c with
 * |c| up to the threshold) enter the residual of a Ritz VECTOR at first order, so the vector is only good to ~threshold */
#define DSEA_PRO_DELTA_DEFAULT 1e-10
#define DSEA_SCAL_LZ_FAIL 38  /* scal[38] = 1 when the single-launch Lanczos lost a peer workgroup (timeout) */

namespace dsea {

struct TfimParams {
  int L, L_local;
  int64_t row_offset;
  const double* g_dev;
  double g_const;
  double diag_scale;
};
struct CsrParams {
  int64_t n, nnz;
  const int64_t* rowptr;
  const int32_t* colidx;
  const double* vals;
};
struct SellParams {
  int64_t n, nslices;
  const int64_t* slice_ptr;
  const int32_t* colidx;
  const double* vals;
  // row-partitioned slab (dsea_pop_create_csr; all zero for a one-GPU operator): where x[col] is read from
  //   mode 0: x itself;  1: LOCAL columns in [-hb, n + hb): c < 0 -> halo_lo[c + hb], c >= n -> halo_hi[c - n];
  //   2: GLOBAL columns, read from the all-gathered copy xg
  int mode;
  int64_t hb;
  const double* halo_lo;
  const double* halo_hi;
  const double* xg;
  // 16-bit column deltas (dsea_op_create_sell16): column of element e = colbase[e / 64] + col16[e]; colidx unused (null)
  const int32_t* colbase;
  const uint16_t* col16;
  int xcd;   // 1: XCD-contiguous slice map (dsea_op_set_tuning DSEA_TUNE_SELL_XCD_MAP)
  int nt;    // 1: non-temporal matrix loads (DSEA_TUNE_SELL_NT; 16-bit-column operands)
  int max_width;  // hint: no slice is wider than this many slice columns (0 = unknown; DSEA_TUNE_SELL_MAX_WIDTH)
  int pack2;  // 1: fp64 values and 16-bit deltas packed TWO slice columns to a lane (dsea_op_create_sell16p2)
  // value-coded operand (dsea_op_create_sell16v8): value of element e = vtab[code8[e]] (256 doubles); vals unused (null)
  const uint8_t* code8;
  const double* vtab;
};
struct Stencil3Params {
  int64_t n;
  double coef;
  const double* V;
  const double* halo_lo;
  const double* halo_hi;
};
struct DenseParams {
  int64_t n, lda;
  const double* A;  // row-major n x n
  int transpose;
};
struct TransferParams {
  int D, d;
  const double* B;  // d x D x D row-major: A itself, or its slice-wise transpose (inside the caller's work buffer)
  double* xT;       // D * D      scratch: the transposed input
  double* T;        // d * D * D  scratch: B_k x
  double* Y;        // d * D * D  scratch: (B_k x) B_k^T
  int transpose;
  const double* Bp; // d * Dp * Dp (Dp = D rounded up to a multiple of 64): B zero-padded, in fragment-packed order (dsea_transfer_mfma.hip)
  double* Tp;       // d * Dp * Dp scratch: B_k x, written and read in fragment-packed order
};
struct SymDenseParams {
  int64_t n, lda, npad;
  const void* A;    // row-major, symmetric, fp64 or fp32 (elem); only the upper triangle is read
  double* work;     // nb x npad doubles of per-tile partial results
  int nb;           // 64-row blocks
  int elem;         // bytes per matrix element: 8 or 4
};
enum OpKind { OP_TFIM = 1, OP_CSR = 2, OP_STENCIL3 = 3, OP_SELL = 4, OP_DENSE = 5, OP_TRANSFER = 6, OP_SYMDENSE = 7 };
struct OpDesc {
  OpKind kind;
  int64_t n;
  int tune_tile_log2;  // TFIM: log2 rows of x staged in LDS per block (6..12)
  int tune_csr_group;  // CSR: lanes per row, 0 = automatic
  int tune_sell_unroll;  // SELL: slice-column pairs in flight per lane {0 = automatic, 2, 4, 6, 8}; 1 = the round-5 kernel (A/B)
  TfimParams tfim;
  CsrParams csr;
  Stencil3Params st3;
  SellParams sell;
  DenseParams dense;
  TransferParams transfer;
  SymDenseParams symdense;
};

// how the rows of one vector are cut into wave tiles for the basis-streaming kernels
struct TileGeom {
  int rpl;         // rows per lane (2,4,8,16): a wave tile is 64*rpl rows
  int nw;          // waves launched (<= DSEA_MAX_WAVE_TILES); wave w handles tiles w, w+nw, ...
  int64_t ntiles;  // ceil(n / (64*rpl))
  int pstride;     // stride between the partial rows of two basis vectors (>= nw)
  int split_w;     // 0, or W = waves of one block that share a 128-row tile and split the basis (small n)
  int dots_w;      // split form of the dots pass: waves per block ...
  int dots_nt;     // ... and 128-row sub-tiles per block (1 or 2); partial count = ceil(ntiles / dots_nt)
};

// optional per-launch HIP-event timing of the dominant kernels (bench.py roofline); host objects only
enum ProfKind { PROF_RDOTS = 0, PROF_AXPY = 1, PROF_SPMV = 2, PROF_KINDS = 3 };
struct EventPair {
  hipEvent_t a, b;
  int kind;
};
struct Profiler {
  EventPair* pairs;
  int capacity, used;
  EventPair* next(int kind) {
    if (used >= capacity) return nullptr;
    pairs[used].kind = kind;
    return &pairs[used++];
  }
};

struct Workspace {
  int64_t n, npad;
  int kmax;
  int rpl_override;
  int split_override;  // -1 automatic, 0 off, 4/8/16 forced
  int persist_override;  // persistent single-launch CG: -1 automatic, 0 off, 1/2/4 = row pairs per thread forced
  int lz_persist;        // single-launch Lanczos for small problems: -1 automatic (on where it applies), 0 off
  int arnoldi_optimistic;  // dsea_arnoldi_extend: 1 = the second Gram-Schmidt pass is not enqueued; a step failing the DGKS test records itself
  int last_cg_form;      // which form the last dsea_cg_run took (DSEA_CG_FORM_*)
  int lose_peer;         // TEST HOOK (dsea_ws_set_fault_injection): the last workgroup of a persistent launch exits at once
  int reorth_passes;     // Gram-Schmidt passes per Lanczos step: 1 (the reference, Lanczos.py:66) or 2 (CGS2 option)
  int partial_reorth;    // 1 = re-orthogonalise only when the omega recurrence says so (option; dsea_ws_set_partial_reorth)
  double pro_delta;      // its threshold on the estimated |q_i . q_k| (DSEA_PRO_DELTA_DEFAULT)
  double* partials;  // DSEA_MAX_WAVE_TILES * max(kmax,1) doubles (also >= DSEA_MAX_EW_BLOCKS)
  double* aux;       // 4 * DSEA_MAX_WAVE_TILES doubles: small partial buffers that must not alias `partials`
  double* coef;      // kmax doubles
  double* coef2;     // second coefficient vector (Arnoldi: DGKS second pass)
  double* zero;      // one device double that is always 0
  double* scal;      // DSEA_SCALARS doubles
  double* vec[4];    // four work vectors of npad doubles
  Profiler* prof;    // null unless dsea_profile_begin was called
  uint16_t* shadow;  // caller-owned bf16 shadow of the basis (k rows x shadow_ld), or null
  int64_t shadow_ld;
  int shadow_rows;
  double lp_tau;     // low-precision pass allowed while max|c_j| <= lp_tau * ||r||
  // row-partitioned library driver: dsea_plz_correct leaves its ||r||^2 partials un-summed (defer_norm) and the NEXT
  // dot-closing call of the step sums both in one launch (k_finalize_pair) -- bit-identical, one launch fewer per step
  int callable_na;       // dsea_lanczos_callable_alpha: number of alpha partials left in aux[0..] for the next callable step
  int defer_norm;
  const double* pend_P;
  int pend_count;
  double* pend_out;
  TileGeom geom(int64_t n_rows) const;
};

void launch_finalize1(const double* P, int count, double* out, hipStream_t st);
int launch_three_term(const double* u, const double* q1, const double* q2, const double* aP, int aCount,
                      double* a_store, const double* beta, double* r, double* P, double* psi, const double* s1,
                      int64_t n, double* brk, hipStream_t st);
void launch_finalize_slot(const double* P, int count, double* out, const double* skip, hipStream_t st);
void launch_rdots(const TileGeom& g, const double* Q, int64_t ldq, int64_t n, int i, const double* u,
                  const double* alpha, const double* beta, double* r, double* P, double* c_out,
                  hipStream_t st, EventPair* ev = nullptr, const double* aP = nullptr, int aCount = 0,
                  double* a_store = nullptr, bool want_rr = false, double* brk = nullptr,
                  const double* sel = nullptr, bool sel_exit = false, const double* uscale = nullptr);
// uscale (u divided by *uscale on the fly) exists for the wave-owned geometry without the partial-reorthogonalisation gate
inline bool rdots_uscale_ok(const TileGeom& g) { return g.split_w == 0; }   // sel: device flag of the partial re-orthogonalisation,
                  // 0 = no basis vectors on this step (three-term update and ||r||^2 only; sel_exit: return at once)
int launch_axpy_norm_lp(int64_t n, int rps, const double* Q, int64_t ldq, const uint16_t* Qs, int64_t lds, int i,
                        const double* c, double tau, double* r, double* P, double* lp_count, hipStream_t st,
                        EventPair* ev = nullptr, const double* brk = nullptr);
int launch_tfim_fused(const OpDesc& op, const double* r, const double* nP, int nCount, double* q_out, double* y,
                      double* beta_store, double* P, hipStream_t st, EventPair* ev = nullptr,
                      uint16_t* qs_out = nullptr, double* brk = nullptr, int step = 0);
int launch_cg_update_fused(double* x, double* r, const double* d, const double* Ad, const double* state,
                           int parity, const double* dP, int dCount, int64_t n, double* P, hipStream_t st);
void launch_cg_direction_fused(const double* r, double* d, double* state, int parity, const double* rP,
                               int rCount, double eps, int64_t n, hipStream_t st);
void launch_axpy_norm(const TileGeom& g, const double* Q, int64_t ldq, int64_t n, int i, const double* c,
                      double* r, double* P, double* nrm2_out, hipStream_t st, EventPair* ev = nullptr,
                      const double* brk = nullptr, const double* sel = nullptr);
void launch_ritz(const TileGeom& g, const double* Q, int64_t ldq, int64_t n, int k, const double* s,
                 double* out, hipStream_t st);
void launch_dot(const double* x, const double* y, int64_t n, double* P, double* out, hipStream_t st);
void launch_pro_update(const double* alphas, const double* betas, const double* rrP, int rrCount, double* rr_store,
                       double* om, int ld, double* flag, double* state, int i, double eps1, double delta,
                       const double* brk, hipStream_t st);
int rdots_partial_count(const TileGeom& g, int i);
void launch_probe(const double* x, double* y, int64_t n, double* P, int nP, hipStream_t st);   // y != null: copy
void launch_shift_dot(const double* x, double* y, const double* shift, const double* skip, int64_t n,
                      double* P, double* out, hipStream_t st);
int launch_shift_dot_partials(const double* x, double* y, const double* shift, const double* skip, int64_t n, double* P,
                              hipStream_t st);
void launch_axpy(double a_host, const double* a_dev, const double* x, double* y, int64_t n, hipStream_t st);
void launch_scale_store(const double* r, const double* nrm2, double* q, double* beta_out, int64_t n,
                        hipStream_t st, uint16_t* qs = nullptr, double* brk = nullptr, int step = 0);
void launch_scale_store_fused(const double* r, const double* nP, int nCount, double* q, uint16_t* qs, double* beta_store,
                              int64_t n, hipStream_t st);
int launch_dot_partials(const double* x, const double* y, int64_t n, double* P, hipStream_t st);
void launch_project_apply(const double* v, const double* a, const double* dot, double* out, int64_t n,
                          hipStream_t st);
void launch_cg_init(const double* b, const double* Ax0, double* r, double* d, double* state, int64_t n,
                    double* P, hipStream_t st);
void launch_cg_init_check(double* state, double eps, hipStream_t st);
void launch_cg_update(double* x, double* r, const double* d, const double* Ad, double* state, int64_t n,
                      double* P, hipStream_t st);
void launch_cg_check(double* state, double eps, hipStream_t st);
int launch_pcg_update(double* x, double* r, double* p, double* s, const double* w, const double* state, int64_t n,
                      double* P, hipStream_t st);
void launch_pcg_scalars(double* state, const double* pair, double eps, int first, hipStream_t st);
void launch_cg_direction(const double* r, double* d, const double* state, int64_t n, hipStream_t st);
void launch_axpy_multi_dot(double a_host, const double* a_dev, const double* const* xs, int count,
                           const double* shift, const double* skip, const double* x, double* y, int64_t n,
                           double* P, double* dot_out, hipStream_t st, const double* pendP = nullptr, int pendN = 0,
                           double* pendOut = nullptr);
void launch_form_r(const double* u, const double* q1, const double* q2, const double* alpha, const double* beta,
                   double* r, double* r_copy, int64_t n, hipStream_t st);
void launch_hypercube_flipsum(const double* xT, double* zT, int P, int p, int64_t chunk, hipStream_t st);
void launch_plz_finish(const double* r, const double* y, const double* pair, double* q, uint16_t* qs, double* u,
                       double* alpha_out, double* beta_out, int64_t n, hipStream_t st);
void launch_plz_finish_form(double* r, const double* y, const double* pair, double* q, uint16_t* qs, const double* qprev,
                            double* alpha_out, double* beta_out, double* r_copy, int64_t n, hipStream_t st);
void launch_finalize_pair(const double* PA, int na, double* outA, const double* PB, int nb, double* outB,
                          const double* skipB, hipStream_t st);
// dsea_krylov.hip
bool blas_available();
int blas_apply(const OpDesc& op, const double* x, double* y, hipStream_t st);
void launch_transpose_sq(const double* in, double* out, int D, int batch, hipStream_t st);
void arnoldi_orth(Workspace& w, int64_t n, const double* u, const double* shift_or_zero, double* V, int64_t ldv, int j,
                  double* hcol, double* brk, double* skip, double* nrm1, double* nrm2, hipStream_t st,
                  bool optimistic = false);
int arnoldi_step(const OpDesc& op, Workspace& w, const double* shift_or_zero, double* V, int64_t ldv, int j,
                 double* hcol, double* brk, double* skip, double* nrm1, double* nrm2, hipStream_t st,
                 bool optimistic = false);
void launch_residual(const double* b, const double* u, double* r, int64_t n, double* P, double* nrm2_out,
                     hipStream_t st);
void launch_gmres_begin(const double* nrm2, double target, double* g, int m, double* state, double* brk,
                        hipStream_t st);
void launch_gmres_givens(double* H, int ldh, int j, double* cs, double* sn, double* g, double target, double* state,
                         double* brk, hipStream_t st);
void launch_gmres_solve(const double* H, int ldh, int m, const double* g, const double* state, double* y,
                        hipStream_t st);
size_t persist_comm_bytes(int64_t n);
int launch_cg_persist(const OpDesc& op, const double* shift, const double* b, double* x, double* state, double eps,
                      int64_t maxiter, void* comm, int ppt_override, hipStream_t st, int lose_peer = 0);
int launch_spmv(const OpDesc& op, const double* x, double* y, const double* shift, const double* skip,
                double* P, hipStream_t st, EventPair* ev = nullptr);
int launch_sell_update_vals(const OpDesc& op, const int64_t* rowptr, const double* vals_csr, hipStream_t st);
int launch_sddmm(const OpDesc& op, const int64_t* rowptr, const double* v1, const double* v2, double alpha, int accumulate,
                 bool sym, double* out, hipStream_t st);
// dsea_cg_persist_tfim_big.hip
bool cg_persist_tfim_big_applicable(const OpDesc& op);
size_t cg_persist_tfim_big_comm_bytes(int64_t n);
int launch_cg_persist_tfim_big(const OpDesc& op, const double* shift, const double* b, double* x, double* state,
                               double eps, int64_t maxiter, void* comm, double* dbuf0, double* dbuf1, hipStream_t st,
                               int lose_peer = 0, bool merged = true);
// dsea_cg_persist_tfim.hip
bool cg_persist_tfim_applicable(const OpDesc& op);
size_t cg_persist_tfim_comm_bytes(int64_t n);
int launch_cg_persist_tfim(const OpDesc& op, const double* shift, const double* b, double* x, double* state, double eps,
                           int64_t maxiter, void* comm, hipStream_t st, int lose_peer = 0);
// dsea_lanczos_persist.hip
bool lanczos_persist_applicable(const OpDesc& op, int64_t n, int k);
size_t lanczos_persist_comm_bytes(int64_t n, int k);
int launch_lanczos_persist(const OpDesc& op, int k, const double* q0, double* Q, int64_t ldq, double* alphas,
                           double* betas, double* brk, double* fail, void* comm, hipStream_t st, int lose_peer = 0);

// dsea_transfer_mfma.hip
bool transfer_mfma_applicable(const OpDesc& op);
int launch_transfer_mfma(const OpDesc& op, const double* x, double* y, hipStream_t st);
void launch_pack_fragments(const double* B, double* Bp, int D, int d, hipStream_t st);
// dsea_lanczos_persist_mid.hip
bool lanczos_persist_mid_applicable(const OpDesc& op, int64_t n, int k);
size_t lanczos_persist_mid_comm_bytes(int64_t n, int k);
int launch_lanczos_persist_mid(const OpDesc& op, int k, const double* q0, double* Q, int64_t ldq, uint16_t* Qs, int64_t lds,
                               double tau, double* alphas, double* betas, double* brk, double* fail, double* lp_count,
                               void* comm, hipStream_t st, int lose_peer = 0);

}  // namespace dsea

// Host-side polling of the CG state without draining the stream (dsea_capi.hip): the state of chunk j is copied into
// pinned memory behind an event while chunk j + 1 is already enqueued -- the device never waits for the host round trip
// (the launches of a chunk issued after convergence are no-ops on the device: every CG kernel tests the DONE flag).
namespace dsea {
struct StatePoller {
  double* pinned;        // 2 x DSEA_CG_STATE_LEN doubles of page-locked host memory
  hipEvent_t ev[2];
  bool ok;
};
StatePoller* state_poller();   // per host thread, created on first use, nullptr if the runtime refuses
}  // namespace dsea

// the opaque handles of include/dsea.h
struct dsea_op_s {
  dsea::OpDesc d;
};
struct dsea_ws_s {
  dsea::Workspace w;
};
#endif
