/*
 * dsea.h -- C ABI of the MI355X-native dominant-eigenpair hot path (libdsea.so).
 *
 * The reference (buwantaiji/DominantSparseEigenAD) is pure Python on torch ops and
 * has no FFI of its own; these entry points are the boundary a maintainer would
 * bind (ctypes stub shown in INTEGRATION.md) to replace, op for op, the torch
 * calls inside the two hot loops:
 *
 *      Lanczos tridiagonalisation      reference DominantSparseEigenAD/Lanczos.py:49-77
 *      Ritz vector                     reference DominantSparseEigenAD/Lanczos.py:98-105
 *      conjugate gradients             reference DominantSparseEigenAD/CG.py:24-41
 *      projections b - (a.b) a         reference CG.py:59,67,122,132 ; symeig.py:27,80
 *      user mat-vec A(v)               reference examples/TFIM/TFIM.py:91-98,
 *                                      examples/schrodinger1D.py:18-27 (+ generic CSR)
 *
 * Conventions
 *   - every function returns an int status: 0 = ok, negative = DSEA_ERR_*; nothing throws;
 *   - all data pointers are DEVICE pointers owned by the caller (PyTorch); the library
 *     allocates no device memory.  The only library-owned objects are the small host-side
 *     handles (dsea_op_t, dsea_ws_t) that carve / describe caller memory;
 *   - vectors are fp64, contiguous, 16-byte aligned (DSEA_ERR_ALIGN otherwise);
 *   - the Krylov basis is stored VECTOR-CONTIGUOUS: vector j starts at Q + j*ldq
 *     (ldq >= n, ldq even).  The reference keeps it (n,k) row-major (Lanczos.py:49);
 *     the host layer hands users the transposed view;
 *   - `stream` is a hipStream_t passed as void*; every function is asynchronous on that
 *     stream unless its comment says it synchronises;
 *   - scalars that feed the next kernel (alpha, beta, CG state) stay on the device: they are
 *     passed as device pointers so that no host round trip sits inside the loops.  In the
 *     row-partitioned multi-GPU mode the caller all-reduces exactly those device scalars
 *     between two phase calls (each phase leaves the LOCAL sum there).
 *   - one workspace per stream; a workspace is not re-entrant; distinct workspaces are
 *     independent.
 */
#ifndef DSEA_H
#define DSEA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSEA_OK 0
#define DSEA_ERR_ARG (-1)       /* null pointer / bad size / bad enum            */
#define DSEA_ERR_ALIGN (-2)     /* pointer not 16-byte aligned or ldq odd        */
#define DSEA_ERR_WORKSPACE (-3) /* workspace too small for (n, k)                */
#define DSEA_ERR_HIP (-4)       /* a HIP runtime call failed (see dsea_last_hip_error) */
#define DSEA_ERR_NOT_CONVERGED (-5) /* dsea_cg_run hit maxiter                   */
#define DSEA_ERR_UNSUPPORTED (-6)
#define DSEA_ERR_TIMEOUT (-8)   /* persistent single-launch solver: a peer workgroup did not arrive (bounded spin) */
#define DSEA_ERR_BREAKDOWN (-7) /* dsea_lanczos_status: the run met beta ~ 0 and stopped itself */

typedef struct dsea_op_s *dsea_op_t; /* operator descriptor (host struct, device pointers inside) */
typedef struct dsea_ws_s *dsea_ws_t; /* workspace descriptor                                     */

int dsea_version(void);
const char *dsea_error_string(int status);
int dsea_last_hip_error(void);

/* ------------------------------------------------------------------ workspace
 * Scratch for partial sums, reorthogonalisation coefficients and four work vectors.
 * dsea_ws_bytes tells the caller how much device memory to provide.  kmax (Krylov vectors the
 * workspace can serve) is capped at 8000: DSEA_ERR_ARG beyond.                                  */
int dsea_ws_bytes(int64_t n, int kmax, size_t *bytes);
int dsea_ws_create(void *device_buffer, size_t bytes, int64_t n, int kmax, dsea_ws_t *out);
int dsea_ws_destroy(dsea_ws_t ws);
/* tuning knob (0 = automatic): rows handled per lane in the basis-streaming kernels {2,4,8,16} */
int dsea_ws_set_rows_per_lane(dsea_ws_t ws, int rpl);
/* tuning knob: small-n "split" form of the basis-streaming kernels -- a block of `waves` waves shares one
 * 128-row tile and splits the basis vectors between its waves.  -1 = automatic (on below ~1.3e5 rows),
 * 0 = off, 4 / 8 / 16 = forced.                                                                          */
int dsea_ws_set_split(dsea_ws_t ws, int waves);
/* Persistent single-launch CG of dsea_cg_run (3-point stencil without halo pointers up to 2^19 rows; full-space matrix-free
 * TFIM at 2^11 ... 2^20 rows): the whole solve is ONE launch.                    design: docs/design/04-kernels.md 3a, 3c, 3e
 *   -1  automatic (on where it applies)          0  off (streaming form: three launches per iteration)
 *   1, 2 / 21, 22 / 11, 12   forced geometry: row pairs per thread in workgroups of 1024 / 512 / 256 threads (stencil)
 *   100 + g   MERGED-REDUCTION form, one grid-wide exchange per iteration (Chronopoulos-Gear recurrences: the same iteration
 *             in exact arithmetic, NOT the rounding sequence of reference CG.py:31-40); never automatic on the stencil
 *   200       TFIM: the TWO-exchange form, iterates bit-identical to the streaming kernels (other operands: as -1)
 * Defaults: stencil -- the reference's recurrences (bit-identical to streaming); TFIM -- the one-exchange form.          */
int dsea_ws_set_persist(dsea_ws_t ws, int mode);
/* which form the LAST dsea_cg_run on this workspace took (a persistent launch that times out is repeated by the caller in
 * the streaming form, whose rounding sequence differs from the one-exchange form's: callers and tests can tell)       */
#define DSEA_CG_FORM_STREAMING 0            /* mat-vec + update + direction launches: CG.py:31-40 in fp64            */
#define DSEA_CG_FORM_PERSISTENT 1           /* one launch, the reference's recurrences (bit-identical to streaming)   */
#define DSEA_CG_FORM_PERSISTENT_MERGED 2    /* one launch, Chronopoulos-Gear recurrences (one exchange per iteration) */
int dsea_cg_last_form(dsea_ws_t ws, int *form);
/* Single-launch Lanczos of dsea_lanczos_run: README-sized problems (TFIM L <= 13, stencil n <= 8192, k <= 512) and the
 * MID-SIZE stencil form (8192 < n <= 131072 rows, k <= 505: BASELINE configs[2]).  Same algorithm and expressions as the
 * multi-launch kernels; partial sums are combined per slab, so T agrees to rounding, not bit for bit.
 *   -1  automatic (on where it is measured to win)    0  off    1  on wherever it applies    2  README-sized form only
 * A lost peer workgroup (bounded spins) makes dsea_lanczos_status return DSEA_ERR_TIMEOUT; the caller repeats the run with
 * the knob off.                                                                  design: docs/design/04-kernels.md 3c, 3d  */
int dsea_ws_set_lanczos_persist(dsea_ws_t ws, int mode);
/* Gram-Schmidt passes per step of dsea_lanczos_run: 1 = the reference (single-pass classical Gram-Schmidt against all
 * previous vectors, Lanczos.py:66), 2 = the pass is repeated on the corrected vector ("CGS2": orthogonality at rounding
 * level even where one pass leaves eps ||u|| / beta) -- an option the reference lacks, never selected automatically. */
int dsea_ws_set_reorth_passes(dsea_ws_t ws, int passes);
/* PARTIAL re-orthogonalisation for dsea_lanczos_run / dsea_pop_lanczos_run (Simon 1984) -- an option the reference lacks
 * (it re-orthogonalises on every step, Lanczos.py:66), never selected automatically.  A step is re-orthogonalised only when
 * the omega recurrence exceeds `delta` (0 = the default 1e-10).  Operators with a fused tail only, multi-launch form, fp64
 * basis; otherwise DSEA_ERR_UNSUPPORTED.  dsea_lanczos_reorth_stats (synchronises): steps re-orthogonalised in the last run
 * and the ||A|| estimate used.                                                  design: docs/design/09-next-rows-f1-f4.md 8.4 */
int dsea_ws_set_partial_reorth(dsea_ws_t ws, int on, double delta);
int dsea_lanczos_reorth_stats(dsea_ws_t ws, int64_t *reorth_steps, double *anorm, void *stream);
/* TEST HOOK for the persistent single-launch forms (Lanczos, TFIM CG): with lose_peer != 0 the last workgroup of such a
 * launch exits at once, so its peers run into their bounded spins -- exercises the DSEA_ERR_TIMEOUT path and the
 * host's fall-back to the multi-launch kernels (a launch then takes the 3 s of the timeout).                      */
int dsea_ws_set_fault_injection(dsea_ws_t ws, int lose_peer);

/* Optional bf16 SHADOW of the Krylov basis (caller-owned, `rows` x `ld` uint16, ld % 8 == 0, 16-byte
 * aligned; null = off).  When registered, dsea_lanczos_run also stores every new basis vector rounded to
 * bf16 and the correction pass  r -= sum_j c_j q_j  (second half of Lanczos.py:66) streams the shadow
 * (2 bytes/element) instead of the fp64 basis.  Arithmetic stays fp64 and the dots pass always reads the
 * fp64 basis.  This is exact to working precision because with full re-orthogonalisation
 * max|c_j| ~ 1e-16..1e-15 ||r|| (the correction lives in the last bit of r); the kernel verifies
 * max|c_j| <= tau ||r|| on the device each step and otherwise reads the fp64 basis.
 * dsea_lanczos_lp_stats (synchronises) reports how many steps of the last run took each path.       */
int dsea_ws_set_shadow(dsea_ws_t ws, void *shadow_bf16, int64_t ld, int rows, double tau);
int dsea_lanczos_lp_stats(dsea_ws_t ws, int64_t *lp_steps, int64_t *fp64_steps, void *stream);

/* Per-launch timing of the dominant kernels with HIP events recorded on the launch stream
 * (measurement aid for bench.py's roofline figure; not part of the numerical path).
 * After dsea_profile_begin, dsea_lanczos_run brackets every launch of kind
 *   0 = re-orthogonalisation dots kernel, 1 = re-orthogonalisation axpy kernel, 2 = operator mat-vec
 * with an event pair; dsea_profile_end synchronises and returns launches[3], total_ms[3] (host). */
int dsea_profile_begin(dsea_ws_t ws, int max_records);
int dsea_profile_end(dsea_ws_t ws, int64_t *launches, double *total_ms);

/* ------------------------------------------------------------------ operators
 * An operator is y = A x on a slab of n_local rows.                                   */

/* Transverse-field Ising chain, matrix-free (replaces the gather-table mat-vec of
 * reference examples/TFIM/TFIM.py:39-51,91-98):
 *     y[i] = diag_scale * d(gi) * x[i]  -  g * sum_{j < L_local} x[i ^ (1<<j)],
 *     gi = row_offset + i,  d(gi) = -(L - 2*popcount(gi ^ rotl_L(gi,1))).
 * g is read from the device pointer g_dev when it is non-null (the parameter tensor of the
 * model, TFIM.py / E0.py:95-96), otherwise g_const is used.  diag_scale = 0, g_const = 1 gives
 * dH/dg of TFIM.py:58-65.  Flips of bits >= L_local (row-partitioned runs) are the caller's
 * exchange step.  n_local = 2^L_local.                                                        */
int dsea_op_create_tfim(int L, int L_local, int64_t row_offset, const double *g_dev, double g_const,
                        double diag_scale, dsea_op_t *out);

/* CSR, caller-owned device arrays: rowptr int64 [n+1], colidx int32 [nnz], vals fp64 [nnz]. */
int dsea_op_create_csr(int64_t n, int64_t nnz, const int64_t *rowptr, const int32_t *colidx,
                       const double *vals, dsea_op_t *out);

/* Sliced ELLPACK with slices of 64 rows (SELL-64), caller-owned device arrays: slice_ptr int64 [nslices+1]
 * (element offsets), and per slice s, column-major blocks: entry k of row 64 s + l at slice_ptr[s] + 64 k + l,
 * padded per slice to its longest row with (colidx = any valid column, vals = 0).  nslices = ceil(n/64).
 * This is the layout the CSR operand of the host layer is converted to: coalesced matrix loads.           */
int dsea_op_create_sell(int64_t n, int64_t nslices, const int64_t *slice_ptr, const int32_t *colidx,
                        const double *vals, dsea_op_t *out);

/* The same layout with 16-BIT COLUMN DELTAS: the column of element e is colbase[e / 64] + coldelta[e] -- one int32 base per
 * 64-element slice column (the smallest column index in it) and a uint16 per element, 10.06 instead of 12 bytes per
 * non-zero.  Valid when every slice column spans fewer than 65536 columns (banded / structured patterns: the 21-nnz/row
 * TFIM matrix does; the host layer checks and falls back to dsea_op_create_sell).  The SELL mat-vec sits on the bytes it
 * moves (docs/design/12-round6.md): fewer bytes is the one lever it has.  colbase: int32 [slice_ptr[nslices] / 64].       */
int dsea_op_create_sell16(int64_t n, int64_t nslices, const int64_t *slice_ptr, const int32_t *colbase,
                          const uint16_t *coldelta, const double *vals, dsea_op_t *out);

/* dsea_op_create_sell16 with the per-element arrays PACKED TWO slice columns to a lane: every slice is padded to an EVEN
 * number of slice columns (slice_ptr counts the padded elements, multiples of 128) and element (slice column 2 G + j, lane l)
 * of the slice starting at slice_ptr[s] sits at slice_ptr[s] + 128 G + 2 l + j of vals[] and coldelta[]; colbase as before.
 * A lane then reads its two values as one 16-byte load and its two deltas as one uint32 -- 4 instead of 6 memory
 * instructions per two non-zeros of a row (beside the matrix stream the kernel sits on the CU's per-lane load rate,
 * docs/design/12-round6.md section 12.9).  Same products in the same order; everything dsea_op_create_sell16 supports
 * (dsea_op_update_vals, dsea_op_sddmm, dsea_op_set_slab) works on it.                                                        */
int dsea_op_create_sell16p2(int64_t n, int64_t nslices, const int64_t *slice_ptr, const int32_t *colbase,
                            const uint16_t *coldelta, const double *vals, dsea_op_t *out);

/* VALUE-CODED SELL-64 for operands whose stored entries take few distinct values (lattice Hamiltonians held as explicit
 * matrices: couplings, fields and a handful of diagonal levels -- the 21-nnz/row TFIM matrix at L = 20 has 11): the value of an
 * element is table256[code], a uint8 per element into 256 doubles (unused entries: anything finite; padding elements code a
 * 0.0), columns as in dsea_op_create_sell16 -- 3.06 instead of 10.06 bytes per non-zero.  The products are formed with the
 * SAME doubles in the same order: the result is bit-identical to dsea_op_create_sell16 on the decoded values.
 * Layout: every slice is padded to a MULTIPLE OF FOUR slice columns (slice_ptr counts the padded elements, multiples of
 * 256), colbase has one entry per slice column as before (a column that is all padding: any valid column), and the two
 * per-element arrays are packed four slice columns to a lane: element (slice column 4 G + j, lane l) of the slice starting
 * at slice_ptr[s] sits at slice_ptr[s] + 256 G + 4 l + j of code[] and coldelta[] -- a lane reads its four codes as one
 * uint32 and its four deltas as one 8-byte word (with 3 bytes per non-zero the kernel sits on the CU's per-lane load rate,
 * not on the fabric: docs/design/12-round6.md).  The table is copied into LDS by every workgroup.
 * Such an operand is READ-ONLY: dsea_op_update_vals and dsea_op_set_slab answer DSEA_ERR_UNSUPPORTED (the host layer
 * re-codes in place, or rebuilds the fp64 layout once the values stop fitting 256 codes); dsea_op_sddmm works.        */
int dsea_op_create_sell16v8(int64_t n, int64_t nslices, const int64_t *slice_ptr, const int32_t *colbase,
                            const uint16_t *coldelta, const uint8_t *code, const double *table256, dsea_op_t *out);

/* 3-point stencil + diagonal (reference examples/schrodinger1D.py:18-27):
 *     y[i] = coef * ((-2 x[i] + x[i+1]) + x[i-1]) + V[i] * x[i],  x[-1] = *halo_lo, x[n] = *halo_hi
 * (null halo pointer = 0, the Dirichlet padding of the reference).                            */
int dsea_op_create_stencil3(int64_t n, double coef, const double *V_dev, const double *halo_lo,
                            const double *halo_hi, dsea_op_t *out);

/* per-operator tuning knobs (measurement aid; nothing process-wide): key DSEA_TUNE_TFIM_TILE_LOG2 = log2 rows of
 * x staged in LDS per block of the TFIM mat-vec (6..12, default 11); DSEA_TUNE_CSR_GROUP = lanes per CSR row
 * (0 = automatic, or 4..64)                                                                                */
#define DSEA_TUNE_TFIM_TILE_LOG2 1
#define DSEA_TUNE_CSR_GROUP 2
/* SELL mat-vec (32-bit columns): slice columns requested per lane before the first gather {0 = automatic (4), 2, 4, 8};
 * 1 = the round-5 kernel kept as the "before" arm of tools/kbench_csr.py                                       */
#define DSEA_TUNE_SELL_UNROLL 3
/* SELL mat-vec: 1 = XCD-contiguous slice map (workgroup b works on eighth b % 8 of the slices), 0 = round-robin (default) */
#define DSEA_TUNE_SELL_XCD_MAP 4
/* SELL mat-vec with 16-bit columns (dsea_op_create_sell16 only; DSEA_ERR_UNSUPPORTED on the packed / value-coded layouts):
 * 1 = non-temporal loads of the matrix stream (values, column deltas), 0 = default policy */
#define DSEA_TUNE_SELL_NT 5
/* SELL operand: a HINT that no slice is wider than `value` slice columns (0 = unknown).  dsea_op_sddmm and dsea_op_update_vals
 * stage a slice's CSR segment in LDS; knowing the widest one they reserve that much instead of 16 KB per wave and more
 * workgroups fit a CU (both are latency-bound).  A wrong hint costs speed, not correctness. */
#define DSEA_TUNE_SELL_MAX_WIDTH 6
int dsea_op_set_tuning(dsea_op_t op, int key, int value);

/* The explicit-matrix operand as a PARAMETER of the primitives.  The reference's contract is that the adjoint of the
 * operator, A-bar = v1 v2^T, is pushed to whatever parameters produced A (reference README.md:88-126, symeig.py:56-64,
 * 82-84; its dense primitive returns exactly that outer product, symeig.py:29).  For a sparse A whose parameters ARE
 * its stored non-zeros the push-forward is the sampled outer product  vals-bar[e] = v1[row(e)] * v2[col(e)].
 * Both calls address the caller's CSR arrays: `rowptr` (int64 [n+1], the array the operator was built from; ignored
 * for a dsea_op_create_csr operand, which has its own) maps entry k of row i of the SELL copy to CSR element
 * rowptr[i] + k -- the order the host layer's conversion keeps.
 *
 * dsea_op_update_vals: the operator's values become vals_csr[0..nnz).  SELL: its value array (caller-owned, handed to
 *     dsea_op_create_sell) is REWRITTEN IN PLACE through that map, padding stays 0; no rebuild of the layout.  CSR: a
 *     device copy into the operator's array unless vals_csr already is that array.
 * dsea_op_sddmm: out[e] = alpha * v1[row(e)] * v2[col(e)]            (flags = 0)
 *     DSEA_SDDMM_ACCUMULATE : out[e] += ...
 *     DSEA_SDDMM_SYMMETRIC  : alpha * (v1[row] v2[col] + v1[col] v2[row]) / 2 -- the adjoint w.r.t. the entries of a matrix
 *                             that is applied as (M + M^T)/2, i.e. of a symmetric operand whose pairs (i,j), (j,i) move together
 *     out: CSR order, nnz doubles.  Deterministic (one writer per element).
 *     On a slab operator (dsea_op_set_slab) v2[col] is read where the mat-vec reads x[col]: the caller has exchanged v2's
 *     halo / gathered copy; DSEA_SDDMM_SYMMETRIC is DSEA_ERR_UNSUPPORTED there (dsea_pop_sddmm does both exchanges).     */
#define DSEA_SDDMM_ACCUMULATE 1
#define DSEA_SDDMM_SYMMETRIC 2
int dsea_op_update_vals(dsea_op_t op, const int64_t *rowptr, const double *vals_csr, void *stream);
int dsea_op_sddmm(dsea_op_t op, const int64_t *rowptr, const double *v1, const double *v2, double alpha, int flags,
                  double *out, void *stream);

/* GEMM-shaped operands of the NON-symmetric primitives (reference eig.py) -- the one place on this path where a matrix core
 * is the right unit.                                                            design: docs/design/09-next-rows-f1-f4.md 8.1
 *   dense   : row-major n x n matrix (eig.py:28-30, DominantEig); transpose != 0 applies A^T.  Hand-written row-streaming
 *             GEMV (HBM-bound); callers that apply A^T repeatedly hand in the transposed matrix.
 *   transfer: MPS transfer matrix of a rank-3 tensor A (d x D x D row-major), dimension D^2, vectors are D x D row-major
 *             (reference examples/TFIM_vumps/general.py:59-66):
 *                 transpose == 0:  y = sum_s A_s x A_s^T        ("Gong",  general.py:59-61)
 *                 transpose != 0:  y = sum_s A_s^T x A_s        ("GongT", general.py:62-64)
 *             Two hand-written fp64 MFMA kernels (csrc/dsea_transfer_mfma.hip; any D, zero-padded to a multiple of 64 inside)
 *             up to D = 768, two rocBLAS GEMMs beyond (bound at run time from the copy already in the process; absent: the
 *             hand-written pair at every size).  DSEA_TRANSFER_MFMA=1 / =0 forces one or the other.
 *             `work`: caller-owned scratch of dsea_op_transfer_work_bytes(D, d); dsea_op_create_transfer fills part of it on
 *             `stream`.  THE TENSOR IS CAPTURED AT CREATION: the hand-written kernels read the fragment-packed copy made
 *             then, the rocBLAS path reads A_dev itself -- A_dev must stay unchanged for the life of the operator, and an
 *             updated tensor needs a new operator.                                                                       */
int dsea_op_create_dense(int64_t n, const double *A_dev, int64_t lda, int transpose, dsea_op_t *out);
/* dense SYMMETRIC operand (reference symeig.py:15-31 DominantSymeig; Lanczos.py:46-49 applies torch.matmul(A, v)):
 * hand-written mat-vec that reads only the UPPER triangle of the row-major matrix -- every 64 x 64 tile is loaded
 * once and used for both its row block and its column block (half the bytes of a GEMV; deterministic, no atomics).
 * The matrix may be fp64 (elem_bytes = 8) or fp32 (elem_bytes = 4: the reference's dense path follows A.dtype,
 * Lanczos.py:47) -- fp32 elements are widened on load, vectors and arithmetic stay fp64, no promoted copy is made.
 * `work`: caller-owned scratch of dsea_op_symdense_work_bytes(n) (per-tile partial results, n^2/8 bytes).  lda even. */
size_t dsea_op_symdense_work_bytes(int64_t n);
int dsea_op_create_symdense(int64_t n, const void *A_dev, int elem_bytes, int64_t lda, double *work, dsea_op_t *out);
size_t dsea_op_transfer_work_bytes(int D, int d);
int dsea_op_create_transfer(int D, int d, const double *A_dev, int transpose, double *work, void *stream,
                            dsea_op_t *out);

/* A SELL operator as the SLAB of a row-partitioned matrix (SURVEY.md 8e "CSR-banded: halo"; no reference counterpart -- the
 * reference is single-device): the operator holds n consecutive rows of a larger matrix and says where x[col] lives when
 * col is not one of its own rows.
 *   halo_width = hb >= 0 : column indices are LOCAL, in [-hb, n + hb): c < 0 reads halo_lo[c + hb] (the last hb elements of the
 *                          previous slab), c >= n reads halo_hi[c - n] (the first hb of the next slab).  A banded matrix
 *                          with half-bandwidth <= hb <= n: one pairwise exchange of hb elements with each neighbour per mat-vec.
 *                          A missing neighbour's pointer may be null (no column points there).
 *   halo_width = -1      : column indices are GLOBAL and x[col] is read from x_gathered (all slabs, rank-major): the
 *                          mat-vec is preceded by an all-gather of x -- the fallback for patterns without a band.
 * The row operand of dsea_spmv (x, y) stays the slab.  dsea_pop_create_csr wraps such an operator with the exchange. */
int dsea_op_set_slab(dsea_op_t op, int64_t halo_width, double *halo_lo, double *halo_hi, double *x_gathered);

int dsea_op_destroy(dsea_op_t op);
int dsea_op_dim(dsea_op_t op, int64_t *n);

/* y = A x - (*shift) x  (shift may be null);  if dot_out != null: *dot_out = x.y (local sum).
 * The Amap(v) of Lanczos.py:54,71 and CG.py:27,31,34,40 (with q.u of Lanczos.py:55,72 / d.Ad of CG.py:31 through dot_out).
 * If skip_flag != null and *skip_flag != 0 the call is a no-op on the device (converged CG). */
int dsea_spmv(dsea_op_t op, dsea_ws_t ws, const double *x, double *y, const double *shift,
              double *dot_out, const double *skip_flag, void *stream);

/* ------------------------------------------------------------------ vector phases (Lanczos)
 * Used one by one in the generic-callable and the multi-GPU modes, and composed by
 * dsea_lanczos_run.                                                                       */

/* out = x.y   (torch.matmul(q, u) of Lanczos.py:55,72 ; the inner products of CG.py:31,37) */
int dsea_dot(dsea_ws_t ws, const double *x, const double *y, int64_t n, double *out, void *stream);

/* y -= (*shift) * x ; *dot_out = x.y      (generic A: turns A(d) into (A - E0) d, CG.py:120)   */
int dsea_shift_dot(dsea_ws_t ws, const double *x, double *y, const double *shift, double *dot_out,
                   const double *skip_flag, int64_t n, void *stream);

/* y += (a_host * (*a_dev)) * x   (a_dev may be null = 1)       (x + alpha d of CG.py:33 for a caller-composed loop) */
int dsea_axpy(dsea_ws_t ws, double a_host, const double *a_dev, const double *x, double *y,
              int64_t n, void *stream);

/* nrm2_out = ||x||^2 (local)   (torch.norm of Lanczos.py:53,69 and CG.py:28,35, before its square root) */
int dsea_nrm2sq(dsea_ws_t ws, const double *x, int64_t n, double *nrm2_out, void *stream);

/* Measurement probe, not part of the path (bench.py "measured_ceilings"; SURVEY.md 8d asks for the box's own streaming
 * ceiling beside the 8 TB/s spec figure): y == NULL streams x[0..n) once through a read-only reduction (per-block sums
 * land in the workspace's partials), otherwise copies x to y; 8 non-temporal 16-byte loads in flight per lane. */
int dsea_probe_stream(dsea_ws_t ws, const double *x, double *y, int64_t n, void *stream);

/* q_out = r / sqrt(*nrm2) ; if beta_out != null: *beta_out = sqrt(*nrm2)   (Lanczos.py:53,69-70,75) */
int dsea_scale_store(dsea_ws_t ws, const double *r, const double *nrm2, double *q_out,
                     double *beta_out, int64_t n, void *stream);

/* Phase 1 of step i (1 <= i < k), Lanczos.py:61 and the first half of :66:
 *     r = u - (*alpha) Q[i-1] - (*beta) Q[i-2]        (beta may be null: 0, the i = 1 case)
 *     c_out[j] = Q[j] . r   for j < i ;  c_out[i] = r . r     (local sums; c_out holds i+1 doubles)   */
int dsea_lanczos_rdots(dsea_ws_t ws, const double *Q, int64_t ldq, int64_t n, int i,
                       const double *u, const double *alpha, const double *beta, double *r,
                       double *c_out, void *stream);

/* Phase 2, second half of Lanczos.py:66 and :69:  r -= sum_j c[j] Q[j] ; *nrm2_out = ||r||^2 (local).
 * If a bf16 shadow of this basis is registered (dsea_ws_set_shadow, rows kept current by dsea_lanczos_store)
 * the pass streams the shadow, subject to the device-side premise max|c_j| <= tau*sqrt(c[i]).            */
int dsea_lanczos_axpy_norm(dsea_ws_t ws, const double *Q, int64_t ldq, int64_t n, int i,
                           const double *c, double *r, double *nrm2_out, void *stream);

/* Phases 1 + 2 of step i under the PARTIAL re-orthogonalisation option (dsea_ws_set_partial_reorth's scheme as a phase call
 * for a caller-supplied mat-vec): r = u - alphas[i-1] Q[i-1] - betas[i-2] Q[i-2]; the omega estimates advance from
 * alphas[0..i-1], betas[0..i-2] (device arrays, as dsea_dot / dsea_lanczos_store leave them) and ||r||; if they ask for it
 * -- decided on the device, no host synchronisation -- r -= Q Q^T r; *nrm2_out = ||r||^2.  i = 1 starts a new run.
 * dsea_lanczos_reorth_stats reports the count.  fp64 basis (a registered bf16 shadow is not used).                  */
int dsea_lanczos_partial_step(dsea_ws_t ws, const double *Q, int64_t ldq, int64_t n, int i, const double *u,
                              const double *alphas, const double *betas, double *r, double *nrm2_out, void *stream);

/* Q[row] = r / sqrt(*nrm2) (and the bf16 shadow row if registered) ; *beta_out = sqrt(*nrm2) (nullable)
 * (Lanczos.py:53,69-70,75 writing into the vector-contiguous basis)                                       */
int dsea_lanczos_store(dsea_ws_t ws, const double *r, const double *nrm2, double *Q, int64_t ldq, int row,
                       double *beta_out, int64_t n, void *stream);

/* One Lanczos step around a CALLER-SUPPLIED mat-vec (reference Lanczos.py:60-75 with Amap a Python function -- the reference's
 * own calling convention, examples/TFIM/E0.py:59-62), as TWO calls per step instead of five phase calls:
 *     dsea_lanczos_callable_alpha(ws, Q[i-1], u, n, NULL)      u = A Q[i-1] is the caller's: leaves the partial sums of
 *                                                              alpha_{i-1} = Q[i-1].u in the workspace (one launch)
 *     dsea_lanczos_callable_step(ws, Q, ldq, n, i, u, alphas, betas, r)
 *          r = u - alpha_{i-1} Q[i-1] - beta_{i-2} Q[i-2] ; c = Q^T r ; r -= Q c ; beta_{i-1} = ||r|| ; Q[i] = r / beta_{i-1}
 *          (and the bf16 shadow row if one is registered); alphas[i-1], betas[i-1] are stored.  Four launches: the two scalar
 *          reductions (alpha, ||r||^2) are summed in the prologues of their consumers, as in dsea_lanczos_run's step.
 * The last alpha has no consumer: pass alpha_out = alphas + (k-1) to the last dsea_lanczos_callable_alpha to have it summed.
 * Always full re-orthogonalisation in one pass (the workspace's partial / CGS2 options apply to dsea_lanczos_run; around a
 * caller's mat-vec they are composed from the phase calls).                                                              */
int dsea_lanczos_callable_alpha(dsea_ws_t ws, const double *q, const double *u, int64_t n, double *alpha_out, void *stream);
int dsea_lanczos_callable_step(dsea_ws_t ws, double *Q, int64_t ldq, int64_t n, int i, const double *u, double *alphas,
                               double *betas, double *r, void *stream);

/* out = sum_{j<k} s[j] Q[j]   (the one needed column of Qk @ eigvecs, Lanczos.py:99-105)      */
int dsea_ritz_combine(dsea_ws_t ws, const double *Q, int64_t ldq, int64_t n, int k,
                      const double *s, double *out, void *stream);

/* out = v - (a.v) a   (CG.py:59,67,122,132 ; symeig.py:27,80); `a_dot_v` (nullable) receives a.v */
int dsea_project_out(dsea_ws_t ws, const double *v, const double *a, double *out, double *a_dot_v,
                     int64_t n, void *stream);

/* ------------------------------------------------------------------ CG phases (CG.py:24-41)
 * state = 8 device doubles: [0] rr  [1] dAd  [2] rr_new  [3] alpha  [4] beta  [5] resnorm
 *                           [6] done (0/1)  [7] iterations                                   */
#define DSEA_CG_RR 0
#define DSEA_CG_DAD 1
#define DSEA_CG_RRNEW 2
#define DSEA_CG_ALPHA 3
#define DSEA_CG_BETA 4
#define DSEA_CG_RESNORM 5
#define DSEA_CG_DONE 6
#define DSEA_CG_ITERS 7
#define DSEA_CG_STATE_LEN 8

/* r = b - Ax0 ; d = r ; state[RR] = r.r (local) ; clears the rest of state               (CG.py:27,30) */
int dsea_cg_init(dsea_ws_t ws, const double *b, const double *Ax0, double *r, double *d,
                 double *state, int64_t n, void *stream);
/* after the caller has made state[RR] global: done = sqrt(rr) < eps                       (CG.py:28-29) */
int dsea_cg_init_check(dsea_ws_t ws, double *state, double eps, void *stream);
/* alpha = rr/dAd ; x += alpha d ; r -= alpha Ad ; state[RRNEW] = r.r (local)              (CG.py:31,33-34) */
int dsea_cg_update(dsea_ws_t ws, double *x, double *r, const double *d, const double *Ad,
                   double *state, int64_t n, void *stream);
/* after state[RRNEW] is global: iterations += 1 ; resnorm = sqrt(rr_new) ; done |= resnorm < eps ;
 * beta = rr_new/rr ; rr = rr_new                                                          (CG.py:35-38) */
int dsea_cg_check(dsea_ws_t ws, double *state, double eps, void *stream);
/* d = r + beta d   (no-op when done)                                                       (CG.py:39) */
int dsea_cg_direction(dsea_ws_t ws, const double *r, double *d, const double *state, int64_t n,
                      void *stream);

/* ONE iteration of CG around a CALLER-SUPPLIED mat-vec (reference CG.py:31-40 with Amap = the user's Python function, the
 * reference's own calling convention): on entry Ad = A d.  Ad -= (*shift) d (shift nullable); alpha = rr / d.Ad;
 * x += alpha d; r -= alpha Ad; the stopping test on ||r||; beta; d = r + beta d -- the three fused launches dsea_cg_run issues
 * per iteration of its streaming form (same kernels, same partial-sum order: bit-identical iterates), in one call instead of
 * dsea_shift_dot + dsea_cg_update + dsea_cg_check + dsea_cg_direction.  `iteration` = 0, 1, 2, ... since dsea_cg_init /
 * dsea_cg_init_check (it selects which of the two rr slots of `state` is current).  No-op on the device once DONE is set. */
int dsea_cg_step(dsea_ws_t ws, double *x, double *r, double *d, double *Ad, const double *shift, double *state, double eps,
                 int64_t iteration, int64_t n, void *stream);

/* ------------------------------------------------------------------ row-partitioned macro phases
 * One Lanczos step of the row-partitioned mode (no reference counterpart) is
 *     dsea_plz_dots            -> caller all-reduces c[0..i]   (i coefficients + ||r||^2)
 *     dsea_plz_correct_matvec  -> caller exchanges r with the partner slabs
 *     dsea_axpy_multi_dot      -> caller all-reduces pair[0..1] = (||r||^2, r.Ar)
 *     dsea_plz_finish
 * i.e. two latency-bound all-reduces and one slab exchange per step.  The mat-vec is applied to the
 * un-normalised r (linearity): beta = sqrt(pair[0]), alpha = pair[1]/pair[0], q = r/beta, u = (A r)/beta.   */

/* r = u - (*alpha) Q[i-1] - (*beta) Q[i-2] (Lanczos.py:61) as a stand-alone pass; r_copy (nullable) receives a
 * second copy -- the snapshot an overlapped slab exchange sends while r is corrected in place.            */
int dsea_lanczos_form_r(dsea_ws_t ws, const double *Q, int64_t ldq, int64_t n, int i, const double *u,
                        const double *alpha, const double *beta, double *r, double *r_copy, void *stream);

/* Top-bit flips of the row-partitioned TFIM mat-vec in TRANSPOSED form (P = 2^p ranks, P >= 4): the caller
 * all-to-alls its slab (chunk c of every rank's slab goes to rank c), calls this on the received buffer
 * xT[P][chunk]:  zT[s][m] = sum_{b<p} xT[s ^ (1<<b)][m],  and all-to-alls zT back (zT[s] to rank s).  Every
 * link then carries 1/P of a slab per phase instead of a whole slab per partner.  (The flips of the top p bits in the
 * gather-table mat-vec of examples/TFIM/TFIM.py:39-51.)                                                   */
int dsea_hypercube_flipsum(const double *xT, double *zT, int P, int64_t chunk, void *stream);

/* r = u - (*alpha) Q[i-1] - (*beta) Q[i-2] ; c_out[j] = Q[j].r (j < i) ; c_out[i] = r.r   (local sums)
 * (Lanczos.py:61 and the inner product of :66, on a slab)                                                 */
int dsea_plz_dots(dsea_ws_t ws, const double *Q, int64_t ldq, int64_t n, int i, const double *u,
                  const double *alpha, const double *beta, double *r, double *c_out, void *stream);
/* row >= 1: r -= sum_{j<row} c[j] Q[j] (bf16 shadow if registered and the premise holds) ; pair_out[0] = ||r||^2
 * (local).  row == 0: only pair_out[0] = r.r.          (the outer product of Lanczos.py:66 and the norm of :69, on a slab) */
int dsea_plz_correct(dsea_ws_t ws, const double *Q, int64_t ldq, int64_t n, int row, const double *c,
                     double *r, double *pair_out, void *stream);
/* the same followed by y = A_local r (Lanczos.py:71 applied to the un-normalised r; operators whose remote part is ADDED after the local one, e.g. TFIM;
 * halo-type operators call dsea_plz_correct, exchange the halo, then dsea_spmv)                            */
int dsea_plz_correct_matvec(dsea_op_t op, dsea_ws_t ws, const double *Q, int64_t ldq, int row,
                            const double *c, double *r, double *y, double *pair_out, void *stream);
/* y += a_host*(*a_dev) * (xs[0] + ... + xs[count-1]) - (*shift) x ; *dot_out = x.y (local).  count <= 6,
 * a_dev / shift / skip_flag nullable.  (remote part of the TFIM mat-vec: a = -g, xs = partner slabs -- the flips of
 * examples/TFIM/TFIM.py:39-51 whose partner row lives on another rank; the dot is Lanczos.py:72 / CG.py:31)   */
int dsea_axpy_multi_dot(dsea_ws_t ws, double a_host, const double *a_dev, const double *const *xs, int count,
                        const double *shift, const double *skip_flag, const double *x, double *y, int64_t n,
                        double *dot_out, void *stream);
/* pair = global (||r||^2, r.Ar): q_out = r/beta (row `row` of the basis; also its bf16 shadow row if registered),
 * u_out = y/beta, *alpha_out = pair[1]/pair[0], *beta_out = beta (nullable)            (Lanczos.py:69-70,72-75) */
int dsea_plz_finish(dsea_ws_t ws, const double *r, const double *y, const double *pair, double *q_out, int row,
                    double *u_out, double *alpha_out, double *beta_out, int64_t n, void *stream);

/* ------------------------------------------------------------------ row-partitioned solvers with the collectives
 * INSIDE the library ("multi-GPU variants taking ncclComm_t / rank / world").
 * No reference counterpart (the reference is single-device).  One process per GPU; every n-vector is this rank's
 * slab of contiguous rows.  The library issues its slab kernels and the RCCL calls back to back on the caller's
 * stream -- no host language between two phases of a step, no host synchronisation inside the Lanczos loop.
 *
 * Communicator.  Two RCCL communicators per rank: one carries the latency-bound all-reduces of the inner products on
 * the solver's stream, the other the bandwidth-bound slab exchange of the mat-vec on a side stream (RCCL orders the
 * operations of ONE communicator even across streams, which would serialise the exchange with the all-reduces it is
 * meant to hide behind).  Either adopt communicators that already exist (ncclComm_t values, e.g. the ones PyTorch's
 * ProcessGroupNCCL holds: torch exposes them) or let the library create its own from two unique ids that the caller
 * distributes (128 bytes each, produced on rank 0).  RCCL is bound at run time from the copy already in the process
 * (DSEA_ERR_UNSUPPORTED if there is none); environment DSEA_RCCL_LIB=<path> binds THAT library instead (RTLD_LOCAL) -- how
 * the tests run this branch with several ranks on one GPU over a stand-in (tests/fake_rccl; docs/design/11-round5.md).  For transports other than RCCL (MPI, gloo in the tests) the caller
 * supplies three blocking callbacks instead; they receive DEVICE pointers and the stream the data was produced on.   */
typedef struct dsea_comm_s *dsea_comm_t;
#define DSEA_COMM_ID_BYTES 128
#define DSEA_ERR_COMM (-9)     /* an RCCL call (or a caller-supplied collective) failed                        */
#define DSEA_ERR_PREMISE (-10) /* overlapped exchange: the premise max|c_j| <= tau ||r|| failed at some step -- results of
                                  this run must be discarded and the run repeated without overlap (identical decision
                                  on every rank: c is replicated)                                               */
int dsea_comm_unique_id(void *id_out /* DSEA_COMM_ID_BYTES */);
int dsea_comm_init_rank(const void *id_coll, const void *id_xchg, int rank, int world, dsea_comm_t *out);
/* adopt existing ncclComm_t values (xchg_comm may equal coll_comm or be null: then exchange and all-reduces share one
 * communicator and are ordered by RCCL); the library never destroys adopted communicators                        */
int dsea_comm_adopt(void *coll_comm, void *xchg_comm, int rank, int world, dsea_comm_t *out);
typedef int (*dsea_allreduce_fn)(void *user, double *buf, int64_t count, void *stream);           /* in-place sum     */
typedef int (*dsea_alltoall_fn)(void *user, const double *send, double *recv, int64_t chunk, void *stream);
typedef int (*dsea_sendrecv_fn)(void *user, const double *send, double *recv, int64_t count, int peer, void *stream);
int dsea_comm_create_callbacks(int rank, int world, dsea_allreduce_fn allreduce, dsea_alltoall_fn alltoall,
                               dsea_sendrecv_fn sendrecv, void *user, dsea_comm_t *out);
int dsea_comm_destroy(dsea_comm_t comm);
/* the collectives themselves, as the solvers issue them (tests, user-side inner products) */
int dsea_comm_allreduce(dsea_comm_t comm, double *buf, int64_t count, void *stream);
int dsea_comm_alltoall(dsea_comm_t comm, const double *send, double *recv, int64_t chunk, void *stream);
/* recv[r * count .. (r+1) * count) = rank r's send[0..count): RCCL -- one group of point-to-point operations on the exchange
 * communicator; callbacks -- world rounds of the pairwise sendrecv callback (round s pairs rank r with (s - r) mod world) */
int dsea_comm_allgather(dsea_comm_t comm, const double *send, double *recv, int64_t count, void *stream);

/* Row-partitioned operator = slab-local operator + communicator + exchange scratch.
 *   tfim     : chain of L sites over world = 2^p ranks, this rank holds rows [rank 2^(L-p), (rank+1) 2^(L-p)).  Low-bit
 *              flips are slab-local (dsea_op_create_tfim with L_local = L - p); the top p bits come from the partner
 *              slabs rank ^ (1<<b): pairwise exchange (p whole slabs) for world = 2, TRANSPOSED form from world = 4 on
 *              (all-to-all, dsea_hypercube_flipsum, all-to-all back: 2 (P-1)/P slabs spread over all links).
 *              `scratch`: caller-owned, dsea_pop_tfim_scratch_doubles(L, world) doubles.  `side_stream` (nullable):
 *              stream for the exchange; with it the exchange runs behind the slab-local work of the step.
 *              flags: DSEA_POP_OVERLAP -- in the Lanczos step the exchange is started on the UN-corrected r before the
 *              dots pass (r - r' = Q c is at the 1e-14 level while |c_j| <~ 1e-15 ||r||); the premise
 *              max|c_j| <= tau ||r|| is checked ON THE DEVICE every step and recorded: dsea_pop_lanczos_status reports
 *              DSEA_ERR_PREMISE after the run instead of a host round trip per step.  In the CG solve the vector is
 *              final when the mat-vec starts, so the side-stream exchange is exact there.
 *              DSEA_POP_PAIRWISE forces the pairwise form at any world size.
 *   stencil3 : dsea_op_create_stencil3 semantics on contiguous slabs; halo2 = two device doubles the exchange fills
 *              (x[-1] from rank-1, x[n] from rank+1; a missing neighbour is the Dirichlet zero).                    */
typedef struct dsea_pop_s *dsea_pop_t;
#define DSEA_POP_OVERLAP 1
#define DSEA_POP_PAIRWISE 2
/* MEASUREMENT ONLY: the slab exchange is not issued and the receive buffers are used as they are -- the numbers are
 * meaningless, the time is that of the same step without its exchange (bench.py: exposed exchange = step - this)  */
#define DSEA_POP_NO_EXCHANGE 4
/* dsea_pop_cg_run: which recurrences.  Default: TFIM -- ONE all-reduce per iteration (Chronopoulos-Gear: r.r and r.A'r
 * reduced together, A'p carried by a recurrence; the same iteration in exact arithmetic, not CG.py:31-40's rounding
 * sequence; what the one-GPU single-launch solver does by default, dsea_ws_set_persist); stencil3 -- the reference's
 * recurrences, two all-reduces per iteration.  The flags force one or the other.                                   */
#define DSEA_POP_CG_REFERENCE 8
#define DSEA_POP_CG_ONE_REDUCTION 16
size_t dsea_pop_tfim_scratch_doubles(int L, int world);
int dsea_pop_create_tfim(int L, dsea_comm_t comm, const double *g_dev, double g_const, double diag_scale,
                         double *scratch, void *side_stream, int flags, double tau, dsea_pop_t *out);
int dsea_pop_create_stencil3(int64_t n_local, double coef, const double *V_dev, double *halo2, dsea_comm_t comm,
                             dsea_pop_t *out);
/*   csr      : explicit sparse matrix in contiguous row slabs of the SAME n_local rows on every rank (the host layer pads
 *              the last slab with empty rows): `local_op` is this rank's SELL operator with its slab description
 *              (dsea_op_set_slab) -- neighbour halo of hb elements per side, or the all-gather fallback.  The pop borrows
 *              local_op's arrays and exchange buffers (the caller keeps them alive) and SNAPSHOTS its descriptor: values
 *              refreshed through dsea_op_update_vals(local_op, ...) are seen (same arrays), a later dsea_op_set_tuning /
 *              dsea_op_set_slab on local_op is not.  Halo-type operator: the exchange
 *              precedes the slab mat-vec (as for stencil3).                                                          */
int dsea_pop_create_csr(dsea_op_t local_op, dsea_comm_t comm, dsea_pop_t *out);
/* the adjoint hook of a row-partitioned explicit matrix (dsea_op_sddmm on slabs): out[e] (+)= alpha v1[row e] v2[col e] for this
 * rank's rows; v2 (and v1 for DSEA_SDDMM_SYMMETRIC) is exchanged like the x of a mat-vec.  rowptr: this slab's CSR row
 * pointers (local element offsets), out: this slab's non-zeros in CSR order.                                          */
int dsea_pop_sddmm(dsea_pop_t pop, const int64_t *rowptr, const double *v1, const double *v2, double alpha, int flags,
                   double *out, void *stream);
int dsea_pop_destroy(dsea_pop_t pop);
int dsea_pop_set_flags(dsea_pop_t pop, int flags);
/* y = (A - (*shift)) x over all ranks (Amap of Lanczos.py:54,71 / CG.py:27,31 and A(v) - E0 v of CG.py:120); if dot_out != null: *dot_out = GLOBAL x.y (all-reduced, identical on every
 * rank); skip_flag as in dsea_spmv.                                                                              */
int dsea_pop_matvec(dsea_pop_t pop, dsea_ws_t ws, const double *x, double *y, const double *shift, double *dot_out,
                    const double *skip_flag, void *stream);
/* out = GLOBAL x.y   (the inner products of Lanczos.py:55,72 and CG.py:31,37 across ranks) */
int dsea_pop_dot(dsea_pop_t pop, dsea_ws_t ws, const double *x, const double *y, int64_t n, double *out, void *stream);
/* k-step Lanczos with full re-orthogonalisation on slabs (reference Lanczos.py:49-77 distributed): per step the
 * macro phases above with TWO all-reduces (coefficients + ||r||^2 ; ||r||^2, r.Ar) and ONE exchange, all issued by
 * the library on `stream` / the side stream.  alphas[k], betas[max(k-1,1)] are replicated bit-identically.  A bf16
 * shadow registered on the workspace is used and kept current.  No host synchronisation inside.                  */
int dsea_pop_lanczos_run(dsea_pop_t pop, dsea_ws_t ws, int k, const double *q0, double *Q, int64_t ldq,
                         double *alphas, double *betas, void *stream);
/* SYNCHRONISES: DSEA_OK, or DSEA_ERR_PREMISE with *step = first step whose overlap premise failed                 */
int dsea_pop_lanczos_status(dsea_pop_t pop, dsea_ws_t ws, int *step, void *stream);
/* CG on (A - (*shift)) x = b on slabs (reference CG.py:24-41 distributed): per iteration one exchange and two scalar
 * all-reduces (reference recurrences) or ONE (DSEA_POP_CG_* above); the stopping test runs on the device on replicated
 * scalars, the host polls every `poll_every` iterations.  SYNCHRONISES before returning.  Work vectors: the workspace's.
 * The one-reduction form carries r AND A'p by recurrences, so its stop is only ACCEPTED after r = b - A'x has been recomputed
 * from x and found below eps (one extra mat-vec + all-reduce per solve); otherwise the recurrences restart from the true
 * residual, and after three such restarts the solve is finished on the reference's recurrences.  resnorm_out then is the
 * TRUE residual norm ||b - A'x||; iters_out counts the iterations of all rounds.                                          */
int dsea_pop_cg_run(dsea_pop_t pop, dsea_ws_t ws, const double *shift, const double *b, double *x, double *state,
                    double eps, int64_t maxiter, int poll_every, int64_t *iters_out, double *resnorm_out, void *stream);

/* ------------------------------------------------------------------ whole solvers (native operator, one GPU)
 * k-step Lanczos with full re-orthogonalisation, entirely on the stream, no host sync.
 *   q0      : start vector (need not be normalised; Lanczos.py:52-53)
 *   Q       : k x ldq basis (output), alphas[k], betas[max(k-1,1)] (outputs, device)            */
int dsea_lanczos_run(dsea_op_t op, dsea_ws_t ws, int k, const double *q0, double *Q, int64_t ldq,
                     double *alphas, double *betas, void *stream);
/* BASIS-FREE Lanczos (an option the reference lacks; it stores all k vectors -- 8 n k bytes, 429 GB at n = 2^28,
 * k = 200 -- and re-orthogonalises against them, Lanczos.py:49,66): the plain three-term recurrence with three
 * rotating vectors Qrot (3 x ldq, caller-owned), NO re-orthogonalisation.  Call once with s = psi = NULL for the
 * tridiagonal (alphas[k], betas[k-1]); after the host has the Ritz coefficients s[k] (device), call again with the
 * same q0: the recurrence is replayed bit-identically and psi = sum_j s[j] q_j is accumulated on the way.  Without
 * re-orthogonalisation converged Ritz values reappear as copies ("ghosts"); the EXTREME eigenpair -- the one the
 * primitives want -- and its residual beta_k |s_k| are unaffected, psi must be normalised by the caller.
 * Operators with a fused Lanczos tail only (matrix-free TFIM, SELL, 3-point stencil).                        */
int dsea_lanczos_run_basisfree(dsea_op_t op, dsea_ws_t ws, int k, const double *q0, double *Qrot, int64_t ldq,
                               double *alphas, double *betas, const double *s, double *psi, void *stream);
/* Breakdown (the reference has no test: Lanczos.py:69-70 divides by beta whatever it is): the run compares every
 * beta_{i-1} ON THE DEVICE with 1e-13 * max_j(|alpha_j|, |beta_j|); at the first one below it records step i,
 * leaves Q[i..], alphas[i..], betas[i..] untouched and turns its remaining launches into no-ops.  This call
 * SYNCHRONISES the stream and returns DSEA_OK, or DSEA_ERR_BREAKDOWN with *break_step = i: the Krylov space of q0
 * has dimension i and the leading i x i block of T holds exact eigenpairs of the operator; DSEA_ERR_TIMEOUT if the
 * single-launch form (dsea_ws_set_lanczos_persist) lost a peer workgroup -- the outputs are then undefined.       */
int dsea_lanczos_status(dsea_ws_t ws, int *break_step, void *stream);

/* ------------------------------------------------------------------ non-symmetric Krylov loops (row f-1)
 * What reference eig.py:29-30,116-117 (ARPACK eigs) and :54-57,137-144 (gmres) do inside SciPy on the host.
 *
 * dsea_arnoldi_extend: Arnoldi factorisation (A - (*shift) I) V_m = V_{m+1} H extended from column j0 to j1
 * (exclusive), entirely on the stream, no host sync:  V = rows of a (>= j1+1) x ldv buffer with V[0..j0] given
 * (orthonormal), H column-major with leading dimension ldh >= j1+1 (columns j0..j1-1 are written in full: h_0..h_j
 * by classical Gram-Schmidt against ALL previous vectors -- so a restarted, non-Hessenberg start block is fine --
 * and h_{j+1,j} = ||w||).  A second Gram-Schmidt pass runs only when the DGKS test asks for it, decided on the
 * device.  An invariant subspace (||w|| <= 1e-13 ||A v_j||) is recorded; dsea_lanczos_status reports the step.  */
int dsea_arnoldi_extend(dsea_op_t op, dsea_ws_t ws, const double *shift, double *V, int64_t ldv, int j0, int j1,
                        double *H, int ldh, void *stream);

/* diagnostics (synchronises): how many steps since the last j0 == 0 call needed the second Gram-Schmidt pass */
int dsea_arnoldi_second_passes(dsea_ws_t ws, int64_t *count, void *stream);

/* OPTIMISTIC second pass (round 4; off by default).  The second Gram-Schmidt pass of dsea_arnoldi_extend is needed rarely
 * (0 of 200 steps on BASELINE configs[3]'s transfer matrix) but its three launches and the two scalar kernels around
 * them are enqueued on every step -- 11 launches, 5 of them returning at once.  With the option on, a step is 6 launches:
 * the DGKS test runs inside the finish kernel, and a step that FAILS it is not finished -- it records itself, every later
 * launch of the run is a no-op, and dsea_arnoldi_status returns DSEA_ERR_SECOND_PASS with *redo_step = that step (the
 * record is cleared by the call).  The caller then repeats the step with the option off -- dsea_arnoldi_extend(..., j, j + 1,
 * ...) -- and continues from j + 1.  Columns of H and vectors of V are bit-identical to the default mode either way.
 * dsea_arnoldi_status SYNCHRONISES: DSEA_OK, DSEA_ERR_BREAKDOWN (*break_step = invariant subspace reached at that step)
 * or DSEA_ERR_SECOND_PASS.  Without the option it reports what dsea_lanczos_status reports for an Arnoldi run.
 * dsea_gmres_cycle / dsea_gmres_step (native operand) honour the option too: a step that fails the test ends the CYCLE
 * early with the columns it has -- a restart, always valid -- and sets state[5] = 1; the caller runs the following
 * cycle(s) with the option off.                                                                                      */
#define DSEA_ERR_SECOND_PASS (-11)
int dsea_ws_set_arnoldi_optimistic(dsea_ws_t ws, int on);
int dsea_arnoldi_status(dsea_ws_t ws, int *break_step, int *redo_step, void *stream);
/* the same record WITHOUT synchronising: enqueues a copy of it into host_record[0] (caller-owned, preferably pinned) on
 * `stream`; the caller waits for its own event and reads: 0 = fine so far, > 0 = invariant subspace at that step, < 0 = step
 * -value - 1 needs its second pass (then dsea_arnoldi_status clears the record as above).  Lets a caller test the stage it
 * has just enqueued while the NEXT stage is already running (krylov.arnoldi_dominant).                                  */
int dsea_arnoldi_status_enqueue(dsea_ws_t ws, double *host_record, void *stream);
/* enqueue (no synchronisation) a clear of the record behind everything issued so far: a caller that stops at a converged
 * stage while a SPECULATIVE stage is still in flight leaves this behind it, so that whatever that stage records (a breakdown,
 * a second-pass request) cannot turn a later continuation (j0 > 0) on the same workspace into no-ops                     */
int dsea_arnoldi_clear_record(dsea_ws_t ws, void *stream);

/* the orthogonalisation of ONE Arnoldi step (one step of what eig.py:29-30,116-117 runs inside ARPACK and :54,57,140,144
 * inside scipy's gmres) when the mat-vec is the caller's code: u = A v_j given, writes column j
 * of H (entries 0..j+1) and V[j+1]; (*shift) v_j is subtracted from u inside the first pass.                    */
int dsea_arnoldi_orth(dsea_ws_t ws, const double *u, const double *shift, double *V, int64_t ldv, int64_t n, int j,
                      double *H, int ldh, void *stream);

/* dsea_gmres_cycle: ONE cycle of restarted GMRES(m) for (A - (*shift) I) x = b (m <= 64): residual, m Arnoldi steps,
 * Givens rotations, back-substitution and the update x += V y -- all on the stream, no host sync inside.  `first`
 * != 0: x is taken as 0 (r0 = b).  `target` = absolute residual bound max(rtol ||b||, atol) (scipy's rule,
 * eig.py:54).  work: dsea_gmres_work_doubles(m) device doubles; V: (m+1) x ldv, zero-initialised once by the caller.
 * state (8 device doubles): [0] residual estimate  [1] converged  [2] columns used  [3] ||r0||  [4] finished early.
 * The caller reads `state` after the cycle (its one sync) and issues the next cycle if [1] == 0.                 */
size_t dsea_gmres_work_doubles(int m);
/* the three stages of a cycle (the gmres calls of eig.py:54,57,140,144), for operands whose mat-vec is the caller's code
 * (op == NULL, u = A v_j supplied per step; Ax = (A - shift) x supplied to begin, NULL for x = 0); dsea_gmres_cycle
 * composes them for native operators                                                                              */
int dsea_gmres_begin(dsea_ws_t ws, const double *b, const double *Ax, double *V, int64_t ldv, int64_t n, int m,
                     double *work, double target, double *state, void *stream);
int dsea_gmres_step(dsea_op_t op, dsea_ws_t ws, const double *shift, const double *u, double *V, int64_t ldv,
                    int64_t n, int j, int m, double *work, double target, double *state, void *stream);
int dsea_gmres_end(dsea_ws_t ws, const double *V, int64_t ldv, int64_t n, int m, double *work, const double *state,
                   double *x, void *stream);
int dsea_gmres_cycle(dsea_op_t op, dsea_ws_t ws, const double *shift, const double *b, double *x, double *V,
                     int64_t ldv, int m, double *work, double target, double *state, int first, void *stream);

/* CG on (A - (*shift) I) x = b from x (in: start vector, out: solution).  Stops when ||r|| < eps
 * (absolute, CG.py:25) or after maxiter iterations.  The loop runs on the device; the host polls the
 * device-side flag every `poll_every` iterations (0 = default).  SYNCHRONISES the stream before
 * returning.  iters_out / resnorm_out are host pointers (nullable).  state = 8 device doubles.    */
int dsea_cg_run(dsea_op_t op, dsea_ws_t ws, const double *shift, const double *b, double *x,
                double *state, double eps, int64_t maxiter, int poll_every, int64_t *iters_out,
                double *resnorm_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DSEA_H */
