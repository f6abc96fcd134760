// dsea_capi.hip -- the extern "C" boundary of libdsea.so (declared in include/dsea.h).
// Host code only: argument checks, workspace carving, kernel sequencing.  No torch types.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include "dsea_internal.h"

using namespace dsea;

namespace {
thread_local int g_last_hip = 0;

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_last_hip = (int)e;
    return DSEA_ERR_HIP;
  }
  return DSEA_OK;
}

inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

struct WsLayout {
  size_t partials_off, aux_off, coef_off, scal_off, vec_off, total;
  int64_t npad;
};
WsLayout ws_layout(int64_t n, int kmax) {
  WsLayout L;
  const int kk = kmax < 1 ? 1 : kmax;
  L.npad = round_up(n < 1 ? 1 : n, 256);
  size_t off = 0;
  L.partials_off = off;
  off += (size_t)DSEA_MAX_WAVE_TILES * (size_t)(kk + 1) * sizeof(double);  // +1 row: ||r||^2 pseudo-vector
  L.aux_off = off;
  off += (size_t)6 * DSEA_MAX_WAVE_TILES * sizeof(double);   // [0,1] alpha / norm partials  [2] axpy_multi_dot  [3] spare  [4,5] omega rows
  L.coef_off = off;
  off += (size_t)2 * round_up(kk + 2, 32) * sizeof(double);   // two coefficient vectors (Arnoldi: DGKS second pass)
  L.scal_off = off;
  off += (size_t)DSEA_SCALARS * sizeof(double);
  off = (size_t)round_up((int64_t)off, 256);
  L.vec_off = off;
  off += 4 * (size_t)L.npad * sizeof(double);
  L.total = off;
  return L;
}
}  // namespace

namespace dsea {
StatePoller* state_poller() {
  // one per host thread AND device: events belong to the device they were created on
  constexpr int MAXDEV = 16;
  static thread_local StatePoller sp[MAXDEV];
  static thread_local bool tried[MAXDEV] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;
  if (!tried[dev]) {
    tried[dev] = true;
    sp[dev].ok = false;
    void* p = nullptr;
    if (hipHostMalloc(&p, 2 * DSEA_CG_STATE_LEN * sizeof(double), hipHostMallocPortable) == hipSuccess &&
        hipEventCreateWithFlags(&sp[dev].ev[0], hipEventDisableTiming) == hipSuccess &&
        hipEventCreateWithFlags(&sp[dev].ev[1], hipEventDisableTiming) == hipSuccess) {
      sp[dev].pinned = static_cast<double*>(p);
      sp[dev].ok = true;
    } else {
      (void)hipGetLastError();
    }
  }
  return sp[dev].ok ? &sp[dev] : nullptr;
}
}  // namespace dsea

TileGeom Workspace::geom(int64_t n_rows) const {
  TileGeom g;
  int rpl = rpl_override;
  if (rpl != 2 && rpl != 4 && rpl != 8 && rpl != 16) {
    // automatic: about one wave per SIMD of the 256 CUs (1024 wave tiles), each with as many rows --
    // i.e. as many independent 16-byte loads in flight and as long contiguous runs per basis vector --
    // as registers allow.  Measured on MI355X at n = 2^20, i = 199 (tools/kbench.py): rpl 2/4/8/16 ->
    // 5.2 / 5.5 / 5.8 / 6.2 TB/s for the axpy pass.
    if (n_rows >= (int64_t)64 * 16 * 1024) rpl = 16;
    else if (n_rows >= (int64_t)64 * 8 * 1024) rpl = 8;
    else if (n_rows >= (int64_t)64 * 4 * 1024) rpl = 4;
    else rpl = 2;
  }
  g.rpl = rpl;
  g.ntiles = (n_rows + 64 * rpl - 1) / (64 * rpl);
  if (g.ntiles < 1) g.ntiles = 1;
  g.nw = (int)(g.ntiles < DSEA_MAX_WAVE_TILES ? g.ntiles : DSEA_MAX_WAVE_TILES);
  g.pstride = DSEA_MAX_WAVE_TILES;
  // small n: a block of W waves per 128-row tile, the basis vectors split between the waves (latency-bound
  // regime).  Measured on MI355X at i = 199 (tools/kbench.py), dots pass one-wave-per-tile -> split:
  // n = 2^12: 27.7 -> 8.2 us, 2^14: 42.3 -> 8.7 us, 2^16: 46.4 -> 26.7 us, 2^17: 51.4 -> 43.9 us, 2^18: no gain.
  g.split_w = 0;
  g.dots_w = 0;
  g.dots_nt = 1;
  const int64_t tiles128 = (n_rows + 127) / 128;
  int sw = split_override;
  if (sw < 0 && rpl_override == 0) sw = tiles128 <= 256 ? 16 : (tiles128 <= 1024 ? 8 : 0);
  if (sw == 4 || sw == 8 || sw == 16) {
    g.split_w = sw;
    g.rpl = 2;
    g.ntiles = tiles128;
    g.nw = (int)tiles128;
    if (tiles128 > DSEA_MAX_WAVE_TILES) g.split_w = 0;  // cannot happen for automatic selection
    // dots pass: beyond 640 tiles (BASELINE config 3: 782) blocks of 16 waves own TWO sub-tiles -- all blocks resident in one
    // round, twice the loads in flight per trip, half the partials (MI355X, n = 1e5: 26.9 -> 23.0 us at i = 150, 49.9 -> 44.2 at
    // i = 299 including the second stage; no gain at 625 tiles and below: profiles/r03_split_dots_subtiles.txt)
    g.dots_w = g.split_w;
    if (split_override < 0 && tiles128 > 640) {
      g.dots_w = 16;
      g.dots_nt = 2;
    }
  }
  return g;
}

extern "C" {

int dsea_version(void) { return 140; }   // 140: fp64-MFMA transfer mat-vec on packed operands, optimistic Arnoldi second pass

const char* dsea_error_string(int status) {
  switch (status) {
    case DSEA_OK: return "ok";
    case DSEA_ERR_ARG: return "invalid argument";
    case DSEA_ERR_ALIGN: return "pointer not 16-byte aligned or odd leading dimension";
    case DSEA_ERR_WORKSPACE: return "workspace too small";
    case DSEA_ERR_HIP: return "HIP runtime error";
    case DSEA_ERR_NOT_CONVERGED: return "CG did not converge within maxiter";
    case DSEA_ERR_UNSUPPORTED: return "unsupported configuration";
    case DSEA_ERR_TIMEOUT: return "a workgroup of a persistent launch did not arrive in time";
    case DSEA_ERR_BREAKDOWN: return "Lanczos breakdown: the Krylov space is smaller than k";
    case DSEA_ERR_COMM: return "a collective (RCCL call or caller-supplied callback) failed";
    case DSEA_ERR_PREMISE: return "overlapped slab exchange: the premise max|c_j| <= tau ||r|| failed at some step";
    case DSEA_ERR_SECOND_PASS: return "optimistic Arnoldi extension: a step needs its second Gram-Schmidt pass (repeat it with the option off)";
    default: return "unknown status";
  }
}

int dsea_last_hip_error(void) { return g_last_hip; }

int dsea_op_set_tuning(dsea_op_t op, int key, int value) {
  if (!op) return DSEA_ERR_ARG;
  switch (key) {
    case DSEA_TUNE_TFIM_TILE_LOG2:
      if (value < 6 || value > 12) return DSEA_ERR_ARG;
      op->d.tune_tile_log2 = value;
      return DSEA_OK;
    case DSEA_TUNE_CSR_GROUP:
      if (value != 0 && value != 4 && value != 8 && value != 16 && value != 32 && value != 64) return DSEA_ERR_ARG;
      op->d.tune_csr_group = value;
      return DSEA_OK;
    case DSEA_TUNE_SELL_UNROLL:
      if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return DSEA_ERR_ARG;
      op->d.tune_sell_unroll = value;
      return DSEA_OK;
    case DSEA_TUNE_SELL_XCD_MAP:
      if (op->d.kind != OP_SELL || (value != 0 && value != 1)) return DSEA_ERR_ARG;
      op->d.sell.xcd = value;
      return DSEA_OK;
    case DSEA_TUNE_SELL_NT:
      if (op->d.kind != OP_SELL || (value != 0 && value != 1)) return DSEA_ERR_ARG;
      // (an A/B switch of the unpacked 16-bit layout: the packed and the value-coded kernels have no such variant)
      if (value && (op->d.sell.pack2 || op->d.sell.code8)) return DSEA_ERR_UNSUPPORTED;
      op->d.sell.nt = value;
      return DSEA_OK;
    case DSEA_TUNE_SELL_MAX_WIDTH:
      if (op->d.kind != OP_SELL || value < 0) return DSEA_ERR_ARG;
      op->d.sell.max_width = value;
      return DSEA_OK;
    default: return DSEA_ERR_ARG;
  }
}

// ---------------------------------------------------------------------------- workspace
int dsea_ws_bytes(int64_t n, int kmax, size_t* bytes) {
  if (!bytes || n < 1 || kmax < 0 || kmax > DSEA_MAX_KRYLOV) return DSEA_ERR_ARG;
  *bytes = ws_layout(n, kmax).total;
  return DSEA_OK;
}

int dsea_ws_create(void* device_buffer, size_t bytes, int64_t n, int kmax, dsea_ws_t* out) {
  if (!device_buffer || !out || n < 1 || kmax < 0 || kmax > DSEA_MAX_KRYLOV) return DSEA_ERR_ARG;
  if (!aligned16(device_buffer)) return DSEA_ERR_ALIGN;
  WsLayout L = ws_layout(n, kmax);
  if (bytes < L.total) return DSEA_ERR_WORKSPACE;
  dsea_ws_s* ws = new (std::nothrow) dsea_ws_s;
  if (!ws) return DSEA_ERR_ARG;
  char* base = static_cast<char*>(device_buffer);
  ws->w.n = n;
  ws->w.npad = L.npad;
  ws->w.kmax = kmax;
  ws->w.rpl_override = 0;
  ws->w.split_override = -1;
  ws->w.persist_override = -1;
  ws->w.lz_persist = -1;
  ws->w.arnoldi_optimistic = 0;
  ws->w.reorth_passes = 1;
  ws->w.partial_reorth = 0;
  ws->w.pro_delta = DSEA_PRO_DELTA_DEFAULT;
  ws->w.lose_peer = 0;
  ws->w.last_cg_form = DSEA_CG_FORM_STREAMING;
  ws->w.prof = nullptr;
  ws->w.shadow = nullptr;
  ws->w.shadow_ld = 0;
  ws->w.shadow_rows = 0;
  ws->w.lp_tau = 1e-12;
  ws->w.callable_na = 0;
  ws->w.defer_norm = 0;
  ws->w.pend_P = nullptr;
  ws->w.pend_count = 0;
  ws->w.pend_out = nullptr;
  ws->w.partials = reinterpret_cast<double*>(base + L.partials_off);
  ws->w.aux = reinterpret_cast<double*>(base + L.aux_off);
  ws->w.coef = reinterpret_cast<double*>(base + L.coef_off);
  ws->w.coef2 = ws->w.coef + round_up((kmax < 1 ? 1 : kmax) + 2, 32);
  ws->w.scal = reinterpret_cast<double*>(base + L.scal_off);
  ws->w.zero = ws->w.scal + 30;
  if (hipMemset(ws->w.scal, 0, DSEA_SCALARS * sizeof(double)) != hipSuccess) {   // scal[30] stays 0 for good
    g_last_hip = (int)hipGetLastError();
    delete ws;
    return DSEA_ERR_HIP;
  }
  for (int v = 0; v < 4; ++v)
    ws->w.vec[v] = reinterpret_cast<double*>(base + L.vec_off) + (size_t)v * (size_t)L.npad;
  *out = ws;
  return DSEA_OK;
}

static void prof_free(Workspace& w) {
  if (!w.prof) return;
  for (int e = 0; e < w.prof->capacity; ++e) {
    (void)hipEventDestroy(w.prof->pairs[e].a);
    (void)hipEventDestroy(w.prof->pairs[e].b);
  }
  delete[] w.prof->pairs;
  delete w.prof;
  w.prof = nullptr;
}

int dsea_ws_destroy(dsea_ws_t ws) {
  if (ws) prof_free(ws->w);
  delete ws;
  return DSEA_OK;
}

int dsea_profile_begin(dsea_ws_t ws, int max_records) {
  if (!ws || max_records < 1) return DSEA_ERR_ARG;
  prof_free(ws->w);
  Profiler* p = new (std::nothrow) Profiler;
  if (!p) return DSEA_ERR_ARG;
  p->pairs = new (std::nothrow) EventPair[max_records];
  p->capacity = max_records;
  p->used = 0;
  for (int e = 0; e < max_records; ++e) {
    if (hipEventCreate(&p->pairs[e].a) != hipSuccess || hipEventCreate(&p->pairs[e].b) != hipSuccess) {
      g_last_hip = (int)hipGetLastError();
      return DSEA_ERR_HIP;
    }
  }
  ws->w.prof = p;
  return DSEA_OK;
}

int dsea_profile_end(dsea_ws_t ws, int64_t* launches, double* total_ms) {
  if (!ws || !ws->w.prof || !launches || !total_ms) return DSEA_ERR_ARG;
  Profiler* p = ws->w.prof;
  for (int kd = 0; kd < PROF_KINDS; ++kd) {
    launches[kd] = 0;
    total_ms[kd] = 0.0;
  }
  for (int e = 0; e < p->used; ++e) {
    if (hipEventSynchronize(p->pairs[e].b) != hipSuccess) return DSEA_ERR_HIP;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p->pairs[e].a, p->pairs[e].b) != hipSuccess) return DSEA_ERR_HIP;
    launches[p->pairs[e].kind] += 1;
    total_ms[p->pairs[e].kind] += (double)ms;
  }
  prof_free(ws->w);
  return DSEA_OK;
}

int dsea_ws_set_shadow(dsea_ws_t ws, void* shadow_bf16, int64_t ld, int rows, double tau) {
  if (!ws) return DSEA_ERR_ARG;
  if (!shadow_bf16) {
    ws->w.shadow = nullptr;
    ws->w.shadow_ld = 0;
    ws->w.shadow_rows = 0;
    return DSEA_OK;
  }
  if (rows < 1 || ld < 8 || tau < 0.0) return DSEA_ERR_ARG;
  if (!aligned16(shadow_bf16) || (ld % 8) != 0) return DSEA_ERR_ALIGN;
  ws->w.shadow = static_cast<uint16_t*>(shadow_bf16);
  ws->w.shadow_ld = ld;
  ws->w.shadow_rows = rows;
  ws->w.lp_tau = tau;
  return DSEA_OK;
}

int dsea_lanczos_lp_stats(dsea_ws_t ws, int64_t* lp_steps, int64_t* fp64_steps, void* stream) {
  if (!ws || !lp_steps || !fp64_steps) return DSEA_ERR_ARG;
  double h[2] = {0.0, 0.0};
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemcpyAsync(h, ws->w.scal + 16, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) {
    g_last_hip = (int)hipGetLastError();
    return DSEA_ERR_HIP;
  }
  *lp_steps = (int64_t)h[0];
  *fp64_steps = (int64_t)h[1];
  return DSEA_OK;
}

int dsea_lanczos_reorth_stats(dsea_ws_t ws, int64_t* reorth_steps, double* anorm, void* stream) {
  if (!ws || !reorth_steps) return DSEA_ERR_ARG;
  double h[2] = {0.0, 0.0};
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemcpyAsync(h, ws->w.scal + DSEA_SCAL_PRO + 2, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) {
    g_last_hip = (int)hipGetLastError();
    return DSEA_ERR_HIP;
  }
  if (anorm) *anorm = h[0];
  *reorth_steps = (int64_t)h[1];
  return DSEA_OK;
}

int dsea_ws_set_rows_per_lane(dsea_ws_t ws, int rpl) {
  if (!ws) return DSEA_ERR_ARG;
  if (rpl != 0 && rpl != 2 && rpl != 4 && rpl != 8 && rpl != 16) return DSEA_ERR_ARG;
  ws->w.rpl_override = rpl;
  return DSEA_OK;
}

int dsea_ws_set_persist(dsea_ws_t ws, int mode) {
  if (!ws) return DSEA_ERR_ARG;
  if (mode == 200) {     // TFIM, 2^11 ... 2^20 rows: the two-exchange persistent form (iterates bit-identical to the streaming
    ws->w.persist_override = mode;   // kernels) instead of the default one-exchange form; other operands: as -1
    return DSEA_OK;
  }
  const int geo = mode >= 100 ? mode - 100 : mode;     // >= 100: merged-reduction form with geometry code mode - 100
  if (geo != -1 && geo != 0 && geo != 1 && geo != 2 && geo != 11 && geo != 12 && geo != 21 && geo != 22)
    return DSEA_ERR_ARG;
  if (mode >= 100 && geo == -1) return DSEA_ERR_ARG;
  ws->w.persist_override = mode;
  return DSEA_OK;
}

int dsea_cg_last_form(dsea_ws_t ws, int* form) {
  if (!ws || !form) return DSEA_ERR_ARG;
  *form = ws->w.last_cg_form;
  return DSEA_OK;
}

int dsea_ws_set_fault_injection(dsea_ws_t ws, int lose_peer) {
  if (!ws) return DSEA_ERR_ARG;
  ws->w.lose_peer = lose_peer ? 1 : 0;
  return DSEA_OK;
}

int dsea_ws_set_reorth_passes(dsea_ws_t ws, int passes) {
  if (!ws || (passes != 1 && passes != 2)) return DSEA_ERR_ARG;
  ws->w.reorth_passes = passes;
  return DSEA_OK;
}

int dsea_ws_set_partial_reorth(dsea_ws_t ws, int on, double delta) {
  if (!ws || !(delta >= 0.0)) return DSEA_ERR_ARG;
  ws->w.partial_reorth = on ? 1 : 0;
  ws->w.pro_delta = delta > 0.0 ? delta : DSEA_PRO_DELTA_DEFAULT;
  return DSEA_OK;
}

int dsea_ws_set_lanczos_persist(dsea_ws_t ws, int mode) {
  // -1 automatic, 0 off, 1 forced wherever a single-launch form applies, 2 = only the README-sized form (the mid-size
  // form of dsea_lanczos_persist_mid.hip off: A/B measurements)
  if (!ws || (mode != -1 && mode != 0 && mode != 1 && mode != 2)) return DSEA_ERR_ARG;
  ws->w.lz_persist = mode;
  return DSEA_OK;
}

namespace {
// geometry of the bf16-shadow correction pass: 0 = split form (16 waves share a 512-row tile and split the basis
// vectors), 1 / 2 = a wave owns 512 / 1024 rows and walks all vectors.  Below 2^20 rows the wave-owned form has
// too few waves to cover the latency of its serial walk (512 waves at n = 2^18 / 2^19): measured per launch (k = 200
// average) n = 2^18: 24.4 -> 17.7 us, 2^19: 31.8 -> 28.1 us with the split form; from 2^20 rows on the wave-owned form
// wins (2^20: 37.0 vs 55.3 us, 2^21: 75.0 vs 99.9 us).
inline int lp_rows_per_step(int64_t n, bool dots_are_split) {
  if (dots_are_split || n < ((int64_t)1 << 20)) return 0;
#ifndef DSEA_LP_RPS
#define DSEA_LP_RPS 2
#endif
  return DSEA_LP_RPS;
}
}  // namespace

int dsea_ws_set_split(dsea_ws_t ws, int waves) {
  if (!ws) return DSEA_ERR_ARG;
  if (waves != -1 && waves != 0 && waves != 4 && waves != 8 && waves != 16) return DSEA_ERR_ARG;
  ws->w.split_override = waves;
  return DSEA_OK;
}

// ---------------------------------------------------------------------------- operators
int dsea_op_create_tfim(int L, int L_local, int64_t row_offset, const double* g_dev, double g_const,
                        double diag_scale, dsea_op_t* out) {
  if (!out || L < 1 || L > 62 || L_local < 0 || L_local > L || row_offset < 0) return DSEA_ERR_ARG;
  if (row_offset & (((int64_t)1 << L_local) - 1)) return DSEA_ERR_ARG;  // slab must be aligned
  dsea_op_s* op = new (std::nothrow) dsea_op_s;
  if (!op) return DSEA_ERR_ARG;
  memset(&op->d, 0, sizeof(op->d));
  op->d.tune_tile_log2 = DSEA_TFIM_TILE_LOG2;
  op->d.kind = OP_TFIM;
  op->d.n = (int64_t)1 << L_local;
  op->d.tfim = TfimParams{L, L_local, row_offset, g_dev, g_const, diag_scale};
  *out = op;
  return DSEA_OK;
}

int dsea_op_create_csr(int64_t n, int64_t nnz, const int64_t* rowptr, const int32_t* colidx,
                       const double* vals, dsea_op_t* out) {
  if (!out || n < 1 || nnz < 0 || !rowptr || (nnz > 0 && (!colidx || !vals))) return DSEA_ERR_ARG;
  dsea_op_s* op = new (std::nothrow) dsea_op_s;
  if (!op) return DSEA_ERR_ARG;
  memset(&op->d, 0, sizeof(op->d));
  op->d.tune_tile_log2 = DSEA_TFIM_TILE_LOG2;
  op->d.kind = OP_CSR;
  op->d.n = n;
  op->d.csr = CsrParams{n, nnz, rowptr, colidx, vals};
  *out = op;
  return DSEA_OK;
}

int dsea_op_create_sell(int64_t n, int64_t nslices, const int64_t* slice_ptr, const int32_t* colidx,
                        const double* vals, dsea_op_t* out) {
  if (!out || n < 1 || nslices != (n + 63) / 64 || !slice_ptr || !colidx || !vals) return DSEA_ERR_ARG;
  dsea_op_s* op = new (std::nothrow) dsea_op_s;
  if (!op) return DSEA_ERR_ARG;
  memset(&op->d, 0, sizeof(op->d));
  op->d.tune_tile_log2 = DSEA_TFIM_TILE_LOG2;
  op->d.kind = OP_SELL;
  op->d.n = n;
  op->d.sell = SellParams{n, nslices, slice_ptr, colidx, vals};
  op->d.sell.xcd = 1;
  *out = op;
  return DSEA_OK;
}

int dsea_op_create_sell16p2(int64_t n, int64_t nslices, const int64_t* slice_ptr, const int32_t* colbase,
                            const uint16_t* coldelta, const double* vals, dsea_op_t* out) {
  const int rc = dsea_op_create_sell16(n, nslices, slice_ptr, colbase, coldelta, vals, out);
  if (rc == DSEA_OK) (*out)->d.sell.pack2 = 1;
  return rc;
}

int dsea_op_create_sell16v8(int64_t n, int64_t nslices, const int64_t* slice_ptr, const int32_t* colbase,
                            const uint16_t* coldelta, const uint8_t* code, const double* table256, dsea_op_t* out) {
  if (!out || n < 1 || nslices != (n + 63) / 64 || !slice_ptr || !colbase || !coldelta || !code || !table256) return DSEA_ERR_ARG;
  dsea_op_s* op = new (std::nothrow) dsea_op_s;
  if (!op) return DSEA_ERR_ARG;
  memset(&op->d, 0, sizeof(op->d));
  op->d.tune_tile_log2 = DSEA_TFIM_TILE_LOG2;
  op->d.kind = OP_SELL;
  op->d.n = n;
  op->d.sell = SellParams{n, nslices, slice_ptr, nullptr, nullptr};
  op->d.sell.colbase = colbase;
  op->d.sell.col16 = coldelta;
  op->d.sell.code8 = code;
  op->d.sell.vtab = table256;
  op->d.sell.xcd = 1;
  *out = op;
  return DSEA_OK;
}

int dsea_op_create_sell16(int64_t n, int64_t nslices, const int64_t* slice_ptr, const int32_t* colbase,
                          const uint16_t* coldelta, const double* vals, dsea_op_t* out) {
  if (!out || n < 1 || nslices != (n + 63) / 64 || !slice_ptr || !colbase || !coldelta || !vals) return DSEA_ERR_ARG;
  dsea_op_s* op = new (std::nothrow) dsea_op_s;
  if (!op) return DSEA_ERR_ARG;
  memset(&op->d, 0, sizeof(op->d));
  op->d.tune_tile_log2 = DSEA_TFIM_TILE_LOG2;
  op->d.kind = OP_SELL;
  op->d.n = n;
  op->d.sell = SellParams{n, nslices, slice_ptr, nullptr, vals};
  op->d.sell.colbase = colbase;
  op->d.sell.col16 = coldelta;
  op->d.sell.xcd = 1;
  *out = op;
  return DSEA_OK;
}

int dsea_op_create_stencil3(int64_t n, double coef, const double* V_dev, const double* halo_lo,
                            const double* halo_hi, dsea_op_t* out) {
  if (!out || n < 1 || !V_dev) return DSEA_ERR_ARG;
  if (!aligned16(V_dev)) return DSEA_ERR_ALIGN;  // read as row pairs (16-byte loads)
  dsea_op_s* op = new (std::nothrow) dsea_op_s;
  if (!op) return DSEA_ERR_ARG;
  memset(&op->d, 0, sizeof(op->d));
  op->d.tune_tile_log2 = DSEA_TFIM_TILE_LOG2;
  op->d.kind = OP_STENCIL3;
  op->d.n = n;
  op->d.st3 = Stencil3Params{n, coef, V_dev, halo_lo, halo_hi};
  *out = op;
  return DSEA_OK;
}

int dsea_op_create_dense(int64_t n, const double* A_dev, int64_t lda, int transpose, dsea_op_t* out) {
  if (!out || n < 1 || !A_dev || lda < n || n > 2147483647ll || lda > 2147483647ll) return DSEA_ERR_ARG;
  dsea_op_s* op = new (std::nothrow) dsea_op_s;
  if (!op) return DSEA_ERR_ARG;
  memset(&op->d, 0, sizeof(op->d));
  op->d.tune_tile_log2 = DSEA_TFIM_TILE_LOG2;
  op->d.kind = OP_DENSE;
  op->d.n = n;
  op->d.dense = DenseParams{n, lda, A_dev, transpose ? 1 : 0};
  *out = op;
  return DSEA_OK;
}

size_t dsea_op_symdense_work_bytes(int64_t n) {
  if (n < 1) return 0;
  const int64_t nb = (n + 63) / 64;
  return (size_t)nb * (size_t)(nb * 64) * sizeof(double);
}

int dsea_op_create_symdense(int64_t n, const void* A_dev, int elem_bytes, int64_t lda, double* work, dsea_op_t* out) {
  if (!out || n < 1 || !A_dev || !work || lda < n || (n + 63) / 64 > 65535 || (elem_bytes != 8 && elem_bytes != 4))
    return DSEA_ERR_ARG;
  if (!aligned16(A_dev) || (lda % 2) != 0) return DSEA_ERR_ALIGN;   // rows are read as element pairs
  dsea_op_s* op = new (std::nothrow) dsea_op_s;
  if (!op) return DSEA_ERR_ARG;
  memset(&op->d, 0, sizeof(op->d));
  op->d.tune_tile_log2 = DSEA_TFIM_TILE_LOG2;
  op->d.kind = OP_SYMDENSE;
  op->d.n = n;
  const int64_t nb = (n + 63) / 64;
  op->d.symdense = SymDenseParams{n, lda, nb * 64, A_dev, work, (int)nb, elem_bytes};
  *out = op;
  return DSEA_OK;
}

size_t dsea_op_transfer_work_bytes(int D, int d) {
  // x^T, T, Y, the slice-wise transposed tensor (D^2 each per slice); the fragment-packed tensor and the packed T of the
  // matrix-core path (Dp^2 each per slice, Dp = D rounded up to a multiple of 64)
  if (D < 1 || d < 1) return 0;
  const size_t Dp = ((size_t)D + 63) / 64 * 64;
  return ((1 + 3 * (size_t)d) * (size_t)D * (size_t)D + 1 + 2 * (size_t)d * Dp * Dp) * sizeof(double);   // (+1: 16-byte alignment of the packed part)
}

int dsea_op_create_transfer(int D, int d, const double* A_dev, int transpose, double* work, void* stream,
                            dsea_op_t* out) {
  if (!out || D < 1 || d < 1 || !A_dev || !work || (int64_t)d * D > 2147483647ll) return DSEA_ERR_ARG;
  if (!aligned16(A_dev) || !aligned16(work)) return DSEA_ERR_ALIGN;
  dsea_op_s* op = new (std::nothrow) dsea_op_s;
  if (!op) return DSEA_ERR_ARG;
  memset(&op->d, 0, sizeof(op->d));
  op->d.tune_tile_log2 = DSEA_TFIM_TILE_LOG2;
  op->d.kind = OP_TRANSFER;
  op->d.n = (int64_t)D * D;
  const size_t DD = (size_t)D * D, slab = (size_t)d * DD;
  double* xT = work;
  double* T = work + DD;
  double* Y = T + slab;
  double* AT = Y + slab;
  const size_t Dp = ((size_t)D + 63) / 64 * 64;
  double* Bp = AT + slab + ((DD + 3 * slab) & 1);      // (16-byte aligned: the packed chunks are read 16 bytes per lane)
  double* Tp = Bp + (size_t)d * Dp * Dp;
  op->d.transfer = TransferParams{D, d, transpose ? AT : A_dev, xT, T, Y, transpose ? 1 : 0, Bp, Tp};
  if (transpose) launch_transpose_sq(A_dev, AT, D, d, static_cast<hipStream_t>(stream));   // B_k = A_k^T, once
  if (Bp) launch_pack_fragments(op->d.transfer.B, Bp, D, d, static_cast<hipStream_t>(stream));   // (MFMA fragment order, once)
  *out = op;
  return check_launch();
}

int dsea_op_update_vals(dsea_op_t op, const int64_t* rowptr, const double* vals_csr, void* stream) {
  if (!op || !vals_csr) return DSEA_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (op->d.kind == OP_SELL) {
    if (!rowptr) return DSEA_ERR_ARG;
    if (op->d.sell.code8) return DSEA_ERR_UNSUPPORTED;      // value-coded operand: read-only (dsea_op_create_sell16v8)
    launch_sell_update_vals(op->d, rowptr, vals_csr, st);
    return check_launch();
  }
  if (op->d.kind == OP_CSR) {
    const CsrParams& p = op->d.csr;
    if (vals_csr != p.vals && p.nnz > 0 &&
        hipMemcpyAsync(const_cast<double*>(p.vals), vals_csr, (size_t)p.nnz * sizeof(double), hipMemcpyDeviceToDevice, st) !=
            hipSuccess)
      return DSEA_ERR_HIP;
    return DSEA_OK;
  }
  return DSEA_ERR_UNSUPPORTED;
}

int dsea_op_sddmm(dsea_op_t op, const int64_t* rowptr, const double* v1, const double* v2, double alpha, int flags,
                  double* out, void* stream) {
  if (!op || !v1 || !v2 || !out || (flags & ~(DSEA_SDDMM_ACCUMULATE | DSEA_SDDMM_SYMMETRIC))) return DSEA_ERR_ARG;
  if (op->d.kind != OP_SELL && op->d.kind != OP_CSR) return DSEA_ERR_UNSUPPORTED;
  if (op->d.kind == OP_SELL && !rowptr) return DSEA_ERR_ARG;
  // a slab (dsea_op_set_slab): the caller has exchanged v2's halo / gathered copy; the symmetric form needs both operands
  // exchanged -- dsea_pop_sddmm issues it as two one-sided launches
  if (op->d.kind == OP_SELL && op->d.sell.mode != 0 && (flags & DSEA_SDDMM_SYMMETRIC)) return DSEA_ERR_UNSUPPORTED;
  if (launch_sddmm(op->d, rowptr, v1, v2, alpha, (flags & DSEA_SDDMM_ACCUMULATE) ? 1 : 0, (flags & DSEA_SDDMM_SYMMETRIC) != 0,
                   out, static_cast<hipStream_t>(stream)) != 0)
    return DSEA_ERR_UNSUPPORTED;
  return check_launch();
}

int dsea_op_set_slab(dsea_op_t op, int64_t halo_width, double* halo_lo, double* halo_hi, double* x_gathered) {
  if (!op || op->d.kind != OP_SELL || op->d.sell.code8) return op ? DSEA_ERR_UNSUPPORTED : DSEA_ERR_ARG;
  SellParams& p = op->d.sell;
  if (halo_width == -1) {
    if (!x_gathered) return DSEA_ERR_ARG;
    p.mode = 2;
    p.hb = 0;
    p.halo_lo = p.halo_hi = nullptr;
    p.xg = x_gathered;
    return DSEA_OK;
  }
  if (halo_width < 0 || halo_width > p.n) return DSEA_ERR_ARG;
  p.mode = 1;
  p.hb = halo_width;
  p.halo_lo = halo_lo;
  p.halo_hi = halo_hi;
  p.xg = nullptr;
  return DSEA_OK;
}

int dsea_op_destroy(dsea_op_t op) {
  delete op;
  return DSEA_OK;
}

int dsea_op_dim(dsea_op_t op, int64_t* n) {
  if (!op || !n) return DSEA_ERR_ARG;
  *n = op->d.n;
  return DSEA_OK;
}

int dsea_spmv(dsea_op_t op, dsea_ws_t ws, const double* x, double* y, const double* shift,
              double* dot_out, const double* skip_flag, void* stream) {
  if (!op || !x || !y || x == y) return DSEA_ERR_ARG;
  if (dot_out && !ws) return DSEA_ERR_ARG;
  if (!aligned16(x) || !aligned16(y)) return DSEA_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  double* P = dot_out ? ws->w.partials : nullptr;
  int nb = launch_spmv(op->d, x, y, shift, skip_flag, P, st);
  if (nb < 0) return DSEA_ERR_UNSUPPORTED;
  if (dot_out) {
    Workspace& w = ws->w;
    if (w.pend_P) {   // the row-partitioned step: the deferred ||r||^2 of dsea_plz_correct is closed in the same launch
      launch_finalize_pair(w.pend_P, w.pend_count, w.pend_out, P, nb, dot_out, skip_flag, st);
      w.pend_P = nullptr;
    } else {
      launch_finalize_slot(P, nb, dot_out, skip_flag, st);
    }
  }
  return check_launch();
}

// ---------------------------------------------------------------------------- vector phases
#define REQUIRE(cond, code) \
  do {                      \
    if (!(cond)) return (code); \
  } while (0)

int dsea_dot(dsea_ws_t ws, const double* x, const double* y, int64_t n, double* out, void* stream) {
  REQUIRE(ws && x && y && out && n >= 1, DSEA_ERR_ARG);
  REQUIRE(aligned16(x) && aligned16(y), DSEA_ERR_ALIGN);
  launch_dot(x, y, n, ws->w.partials, out, static_cast<hipStream_t>(stream));
  return check_launch();
}

int dsea_probe_stream(dsea_ws_t ws, const double* x, double* y, int64_t n, void* stream) {
  REQUIRE(ws && x && n >= 1, DSEA_ERR_ARG);
  REQUIRE(aligned16(x) && (!y || aligned16(y)), DSEA_ERR_ALIGN);
  launch_probe(x, y, n, ws->w.partials, 2048, static_cast<hipStream_t>(stream));
  return check_launch();
}

int dsea_shift_dot(dsea_ws_t ws, const double* x, double* y, const double* shift, double* dot_out,
                   const double* skip_flag, int64_t n, void* stream) {
  REQUIRE(ws && x && y && dot_out && n >= 1, DSEA_ERR_ARG);
  REQUIRE(aligned16(x) && aligned16(y), DSEA_ERR_ALIGN);
  launch_shift_dot(x, y, shift, skip_flag, n, ws->w.partials, dot_out, static_cast<hipStream_t>(stream));
  return check_launch();
}

int dsea_axpy(dsea_ws_t ws, double a_host, const double* a_dev, const double* x, double* y, int64_t n,
              void* stream) {
  (void)ws;
  REQUIRE(x && y && n >= 1, DSEA_ERR_ARG);
  REQUIRE(aligned16(x) && aligned16(y), DSEA_ERR_ALIGN);
  launch_axpy(a_host, a_dev, x, y, n, static_cast<hipStream_t>(stream));
  return check_launch();
}

int dsea_nrm2sq(dsea_ws_t ws, const double* x, int64_t n, double* nrm2_out, void* stream) {
  return dsea_dot(ws, x, x, n, nrm2_out, stream);
}

int dsea_scale_store(dsea_ws_t ws, const double* r, const double* nrm2, double* q_out, double* beta_out,
                     int64_t n, void* stream) {
  (void)ws;
  REQUIRE(r && nrm2 && q_out && n >= 1, DSEA_ERR_ARG);
  REQUIRE(aligned16(r) && aligned16(q_out), DSEA_ERR_ALIGN);
  launch_scale_store(r, nrm2, q_out, beta_out, n, static_cast<hipStream_t>(stream));
  return check_launch();
}

int dsea_lanczos_rdots(dsea_ws_t ws, const double* Q, int64_t ldq, int64_t n, int i, const double* u,
                       const double* alpha, const double* beta, double* r, double* c_out, void* stream) {
  REQUIRE(ws && Q && u && alpha && r && c_out && n >= 1 && i >= 1 && ldq >= n, DSEA_ERR_ARG);
  REQUIRE(i <= ws->w.kmax, DSEA_ERR_WORKSPACE);
  REQUIRE(aligned16(Q) && aligned16(u) && aligned16(r) && (ldq % 2 == 0), DSEA_ERR_ALIGN);
  TileGeom g = ws->w.geom(n);
  Profiler* prof = ws->w.prof;
  launch_rdots(g, Q, ldq, n, i, u, alpha, beta, r, ws->w.partials, c_out, static_cast<hipStream_t>(stream),
               prof ? prof->next(PROF_RDOTS) : nullptr, nullptr, 0, nullptr, true);
  return check_launch();
}

int dsea_lanczos_axpy_norm(dsea_ws_t ws, const double* Q, int64_t ldq, int64_t n, int i, const double* c,
                           double* r, double* nrm2_out, void* stream) {
  REQUIRE(ws && Q && c && r && nrm2_out && n >= 1 && i >= 1 && ldq >= n, DSEA_ERR_ARG);
  REQUIRE(aligned16(Q) && aligned16(r) && (ldq % 2 == 0), DSEA_ERR_ALIGN);
  hipStream_t st = static_cast<hipStream_t>(stream);
  Workspace& w = ws->w;
  if (w.shadow && w.shadow_rows >= i && w.shadow_ld >= n && !w.geom(n).split_w) {
    // a bf16 shadow of this basis is registered: stream it (premise checked on the device against c[i] = r.r)
    double* nP = w.aux + DSEA_MAX_WAVE_TILES;
    const int rps = lp_rows_per_step(n, false);
    int nn = launch_axpy_norm_lp(n, rps, Q, ldq, w.shadow, w.shadow_ld, i, c, w.lp_tau, r, nP, w.scal + 16, st);
    launch_finalize1(nP, nn, nrm2_out, st);
  } else {
    TileGeom g = w.geom(n);
    launch_axpy_norm(g, Q, ldq, n, i, c, r, w.partials, nrm2_out, st);
  }
  return check_launch();
}

int dsea_lanczos_partial_step(dsea_ws_t ws, const double* Q, int64_t ldq, int64_t n, int i, const double* u,
                              const double* alphas, const double* betas, double* r, double* nrm2_out, void* stream) {
  REQUIRE(ws && Q && u && alphas && betas && r && nrm2_out && n >= 1 && i >= 1 && ldq >= n, DSEA_ERR_ARG);
  REQUIRE(i <= ws->w.kmax, DSEA_ERR_WORKSPACE);
  REQUIRE(aligned16(Q) && aligned16(u) && aligned16(r) && (ldq % 2 == 0), DSEA_ERR_ALIGN);
  hipStream_t st = static_cast<hipStream_t>(stream);
  Workspace& w = ws->w;
  REQUIRE(w.n >= n, DSEA_ERR_WORKSPACE);
  const TileGeom g = w.geom(n);
  double* flag = w.scal + DSEA_SCAL_PRO;
  double* om = w.aux + 4 * DSEA_MAX_WAVE_TILES;
  if (i == 1 && hipMemsetAsync(flag, 0, 4 * sizeof(double), st) != hipSuccess) {   // a new run: estimates and counters restart
    g_last_hip = (int)hipGetLastError();
    return DSEA_ERR_HIP;
  }
  // the sequence of dsea_lanczos_run's partial mode, as one phase call around the caller's own mat-vec
  launch_rdots(g, Q, ldq, n, i, u, alphas + (i - 1), i >= 2 ? betas + (i - 2) : nullptr, r, w.partials, nullptr, st, nullptr,
               nullptr, 0, nullptr, true, nullptr, w.zero, false);
  launch_pro_update(alphas, betas, w.partials + (int64_t)i * g.pstride, rdots_partial_count(g, i), w.coef + i, om,
                    DSEA_MAX_WAVE_TILES, flag, flag + 1, i, 64.0 * 2.220446049250313e-16, w.pro_delta, nullptr, st);
  launch_rdots(g, Q, ldq, n, i, r, w.zero, nullptr, w.vec[2], w.partials, w.coef, st, nullptr, nullptr, 0, nullptr, false,
               nullptr, flag, true);
  launch_axpy_norm(g, Q, ldq, n, i, w.coef, r, w.partials, nrm2_out, st, nullptr, nullptr, flag);
  return check_launch();
}

int dsea_lanczos_store(dsea_ws_t ws, const double* r, const double* nrm2, double* Q, int64_t ldq, int row,
                       double* beta_out, int64_t n, void* stream) {
  REQUIRE(ws && r && nrm2 && Q && n >= 1 && row >= 0 && ldq >= n, DSEA_ERR_ARG);
  REQUIRE(aligned16(r) && aligned16(Q) && (ldq % 2 == 0), DSEA_ERR_ALIGN);
  Workspace& w = ws->w;
  uint16_t* qs = nullptr;
  if (w.shadow && w.shadow_rows > row && w.shadow_ld >= n) qs = w.shadow + (int64_t)row * w.shadow_ld;
  launch_scale_store(r, nrm2, Q + (int64_t)row * ldq, beta_out, n, static_cast<hipStream_t>(stream), qs);
  return check_launch();
}

int dsea_lanczos_callable_alpha(dsea_ws_t ws, const double* q, const double* u, int64_t n, double* alpha_out, void* stream) {
  REQUIRE(ws && q && u && n >= 1, DSEA_ERR_ARG);
  REQUIRE(aligned16(q) && aligned16(u), DSEA_ERR_ALIGN);
  Workspace& w = ws->w;
  hipStream_t st = static_cast<hipStream_t>(stream);
  w.callable_na = launch_dot_partials(q, u, n, w.aux, st);                   // consumed by the next dsea_lanczos_callable_step
  if (alpha_out) launch_finalize1(w.aux, w.callable_na, alpha_out, st);      // (the last step: nobody else will sum them)
  return check_launch();
}

int dsea_lanczos_callable_step(dsea_ws_t ws, double* Q, int64_t ldq, int64_t n, int i, const double* u, double* alphas,
                               double* betas, double* r, void* stream) {
  REQUIRE(ws && Q && u && alphas && betas && r && n >= 1 && i >= 1 && ldq >= n, DSEA_ERR_ARG);
  REQUIRE(i <= ws->w.kmax, DSEA_ERR_WORKSPACE);
  REQUIRE(aligned16(Q) && aligned16(u) && aligned16(r) && (ldq % 2 == 0), DSEA_ERR_ALIGN);
  Workspace& w = ws->w;
  REQUIRE(w.callable_na > 0, DSEA_ERR_ARG);                                  // dsea_lanczos_callable_alpha first
  // (always ONE full re-orthogonalisation pass, whatever dsea_ws_set_reorth_passes / _partial_reorth say: those are options of
  //  dsea_lanczos_run; a caller that wants them around its own mat-vec composes them from the phase calls)
  hipStream_t st = static_cast<hipStream_t>(stream);
  const TileGeom g = w.geom(n);
  double* aP = w.aux;
  double* nP = w.aux + DSEA_MAX_WAVE_TILES;
  // (the bf16 shadow of the basis, if registered, under dsea_lanczos_axpy_norm's own condition)
  uint16_t* Qs = (w.shadow && w.shadow_rows >= i && w.shadow_ld >= n && !g.split_w) ? w.shadow : nullptr;
  const int rps = lp_rows_per_step(n, false);
  // the four launches of dsea_lanczos_run's step with the operator's fused tail replaced by a fused normalise-and-store:
  // alpha_{i-1} from the partials the caller's q.u left, ||r||^2 from the correction pass's partials -- no stand-alone
  // second-stage launches (Lanczos.py:61,66,69-70,73-75)
  launch_rdots(g, Q, ldq, n, i, u, nullptr, i >= 2 ? betas + (i - 2) : nullptr, r, w.partials, w.coef, st, nullptr, aP,
               w.callable_na, alphas + (i - 1), Qs != nullptr, nullptr);
  int nn = g.nw;
  if (Qs)
    nn = launch_axpy_norm_lp(n, rps, Q, ldq, Qs, w.shadow_ld, i, w.coef, w.lp_tau, r, nP, w.scal + 16, st);
  else
    launch_axpy_norm(g, Q, ldq, n, i, w.coef, r, nP, nullptr, st);
  uint16_t* qs_row = (w.shadow && w.shadow_rows > i && w.shadow_ld >= n) ? w.shadow + (int64_t)i * w.shadow_ld : nullptr;
  launch_scale_store_fused(r, nP, nn, Q + (int64_t)i * ldq, qs_row, betas + (i - 1), n, st);
  w.callable_na = 0;
  return check_launch();
}

int dsea_ritz_combine(dsea_ws_t ws, const double* Q, int64_t ldq, int64_t n, int k, const double* s,
                      double* out, void* stream) {
  REQUIRE(ws && Q && s && out && n >= 1 && k >= 1 && ldq >= n, DSEA_ERR_ARG);
  REQUIRE(aligned16(Q) && aligned16(out) && (ldq % 2 == 0), DSEA_ERR_ALIGN);
  TileGeom g = ws->w.geom(n);
  launch_ritz(g, Q, ldq, n, k, s, out, static_cast<hipStream_t>(stream));
  return check_launch();
}

int dsea_project_out(dsea_ws_t ws, const double* v, const double* a, double* out, double* a_dot_v,
                     int64_t n, void* stream) {
  REQUIRE(ws && v && a && out && n >= 1, DSEA_ERR_ARG);
  REQUIRE(aligned16(v) && aligned16(a) && aligned16(out), DSEA_ERR_ALIGN);
  hipStream_t st = static_cast<hipStream_t>(stream);
  double* slot = a_dot_v ? a_dot_v : ws->w.scal + 8;
  launch_dot(a, v, n, ws->w.partials, slot, st);
  launch_project_apply(v, a, slot, out, n, st);
  return check_launch();
}

// ---------------------------------------------------------------------------- CG phases
int dsea_cg_init(dsea_ws_t ws, const double* b, const double* Ax0, double* r, double* d, double* state,
                 int64_t n, void* stream) {
  REQUIRE(ws && b && Ax0 && r && d && state && n >= 1, DSEA_ERR_ARG);
  REQUIRE(aligned16(b) && aligned16(Ax0) && aligned16(r) && aligned16(d), DSEA_ERR_ALIGN);
  launch_cg_init(b, Ax0, r, d, state, n, ws->w.partials, static_cast<hipStream_t>(stream));
  return check_launch();
}

int dsea_cg_init_check(dsea_ws_t ws, double* state, double eps, void* stream) {
  (void)ws;
  REQUIRE(state, DSEA_ERR_ARG);
  launch_cg_init_check(state, eps, static_cast<hipStream_t>(stream));
  return check_launch();
}

int dsea_cg_update(dsea_ws_t ws, double* x, double* r, const double* d, const double* Ad, double* state,
                   int64_t n, void* stream) {
  REQUIRE(ws && x && r && d && Ad && state && n >= 1, DSEA_ERR_ARG);
  REQUIRE(aligned16(x) && aligned16(r) && aligned16(d) && aligned16(Ad), DSEA_ERR_ALIGN);
  launch_cg_update(x, r, d, Ad, state, n, ws->w.partials, static_cast<hipStream_t>(stream));
  return check_launch();
}

int dsea_cg_check(dsea_ws_t ws, double* state, double eps, void* stream) {
  (void)ws;
  REQUIRE(state, DSEA_ERR_ARG);
  launch_cg_check(state, eps, static_cast<hipStream_t>(stream));
  return check_launch();
}

int dsea_cg_direction(dsea_ws_t ws, const double* r, double* d, const double* state, int64_t n,
                      void* stream) {
  (void)ws;
  REQUIRE(r && d && state && n >= 1, DSEA_ERR_ARG);
  REQUIRE(aligned16(r) && aligned16(d), DSEA_ERR_ALIGN);
  launch_cg_direction(r, d, state, n, static_cast<hipStream_t>(stream));
  return check_launch();
}

int dsea_cg_step(dsea_ws_t ws, double* x, double* r, double* d, double* Ad, const double* shift, double* state, double eps,
                 int64_t iteration, int64_t n, void* stream) {
  REQUIRE(ws && x && r && d && Ad && state && n >= 1 && iteration >= 0, DSEA_ERR_ARG);
  REQUIRE(aligned16(x) && aligned16(r) && aligned16(d) && aligned16(Ad), DSEA_ERR_ALIGN);
  hipStream_t st = static_cast<hipStream_t>(stream);
  Workspace& w = ws->w;
  double* dP = w.aux;
  double* rP = w.aux + DSEA_MAX_WAVE_TILES;
  const int parity = (int)(iteration & 1);
  // the three launches dsea_cg_run issues per iteration of its streaming form, with the mat-vec replaced by the caller's A d
  const int nd = launch_shift_dot_partials(d, Ad, shift, state + DSEA_CG_DONE, n, dP, st);            // (A - shift) d, d.A'd partials
  const int nr = launch_cg_update_fused(x, r, d, Ad, state, parity, dP, nd, n, rP, st);               // CG.py:31,33-34
  launch_cg_direction_fused(r, d, state, parity, rP, nr, eps, n, st);                                 // CG.py:35-39
  return check_launch();
}


// ---------------------------------------------------------------------------- row-partitioned macro phases
int dsea_axpy_multi_dot(dsea_ws_t ws, double a_host, const double* a_dev, const double* const* xs, int count,
                        const double* shift, const double* skip_flag, const double* x, double* y, int64_t n,
                        double* dot_out, void* stream) {
  REQUIRE(ws && x && y && dot_out && n >= 1 && count >= 0 && count <= 6 && (count == 0 || xs), DSEA_ERR_ARG);
  REQUIRE(aligned16(x) && aligned16(y), DSEA_ERR_ALIGN);
  for (int b = 0; b < count; ++b) {
    REQUIRE(xs[b] != nullptr, DSEA_ERR_ARG);
    REQUIRE(aligned16(xs[b]), DSEA_ERR_ALIGN);
  }
  Workspace& w = ws->w;
  launch_axpy_multi_dot(a_host, a_dev, xs, count, shift, skip_flag, x, y, n, w.aux + 2 * DSEA_MAX_WAVE_TILES,
                        dot_out, static_cast<hipStream_t>(stream), w.pend_P, w.pend_count, w.pend_out);
  w.pend_P = nullptr;
  return check_launch();
}

int dsea_lanczos_form_r(dsea_ws_t ws, const double* Q, int64_t ldq, int64_t n, int i, const double* u,
                        const double* alpha, const double* beta, double* r, double* r_copy, void* stream) {
  (void)ws;
  REQUIRE(Q && u && alpha && r && n >= 1 && i >= 1 && ldq >= n, DSEA_ERR_ARG);
  REQUIRE(aligned16(Q) && aligned16(u) && aligned16(r) && (!r_copy || aligned16(r_copy)) && (ldq % 2 == 0),
          DSEA_ERR_ALIGN);
  const double* q1 = Q + (int64_t)(i - 1) * ldq;
  const double* q2 = (i >= 2) ? Q + (int64_t)(i - 2) * ldq : nullptr;
  launch_form_r(u, q1, q2, alpha, beta, r, r_copy, n, static_cast<hipStream_t>(stream));
  return check_launch();
}

int dsea_hypercube_flipsum(const double* xT, double* zT, int P, int64_t chunk, void* stream) {
  REQUIRE(xT && zT && xT != zT && P >= 1 && chunk >= 1, DSEA_ERR_ARG);
  int p = 0;
  while ((1 << p) < P) ++p;
  REQUIRE((1 << p) == P, DSEA_ERR_ARG);
  launch_hypercube_flipsum(xT, zT, P, p, chunk, static_cast<hipStream_t>(stream));
  return check_launch();
}

int dsea_plz_dots(dsea_ws_t ws, const double* Q, int64_t ldq, int64_t n, int i, const double* u,
                  const double* alpha, const double* beta, double* r, double* c_out, void* stream) {
  REQUIRE(ws && Q && u && alpha && r && c_out && n >= 1 && i >= 1 && ldq >= n, DSEA_ERR_ARG);
  REQUIRE(i <= ws->w.kmax, DSEA_ERR_WORKSPACE);
  REQUIRE(aligned16(Q) && aligned16(u) && aligned16(r) && (ldq % 2 == 0), DSEA_ERR_ALIGN);
  TileGeom g = ws->w.geom(n);
  Profiler* prof = ws->w.prof;
  launch_rdots(g, Q, ldq, n, i, u, alpha, beta, r, ws->w.partials, c_out, static_cast<hipStream_t>(stream),
               prof ? prof->next(PROF_RDOTS) : nullptr, nullptr, 0, nullptr, true);
  return check_launch();
}

int dsea_plz_correct(dsea_ws_t ws, const double* Q, int64_t ldq, int64_t n, int row, const double* c, double* r,
                     double* pair_out, void* stream) {
  REQUIRE(ws && r && pair_out && row >= 0 && n >= 1, DSEA_ERR_ARG);
  REQUIRE(aligned16(r), DSEA_ERR_ALIGN);
  hipStream_t st = static_cast<hipStream_t>(stream);
  Workspace& w = ws->w;
  double* nP = w.aux + DSEA_MAX_WAVE_TILES;
  if (row >= 1) {
    REQUIRE(Q && c && ldq >= n && (ldq % 2 == 0) && aligned16(Q), DSEA_ERR_ARG);
    if (w.shadow && w.shadow_rows > row && w.shadow_ld >= n && !w.geom(n).split_w) {
      const int rps = lp_rows_per_step(n, false);
      int nn = launch_axpy_norm_lp(n, rps, Q, ldq, w.shadow, w.shadow_ld, row, c, w.lp_tau, r, nP, w.scal + 16, st,
                                   w.prof ? w.prof->next(PROF_AXPY) : nullptr);
      if (w.defer_norm) {     // (library driver: summed together with the mat-vec's dot, see Workspace::defer_norm)
        w.pend_P = nP;
        w.pend_count = nn;
        w.pend_out = pair_out;
      } else {
        launch_finalize1(nP, nn, pair_out, st);
      }
    } else {
      TileGeom g = w.geom(n);
      launch_axpy_norm(g, Q, ldq, n, row, c, r, w.partials, pair_out, st, w.prof ? w.prof->next(PROF_AXPY) : nullptr);
    }
  } else {
    launch_dot(r, r, n, w.partials, pair_out, st);
  }
  return check_launch();
}

int dsea_plz_correct_matvec(dsea_op_t op, dsea_ws_t ws, const double* Q, int64_t ldq, int row, const double* c,
                            double* r, double* y, double* pair_out, void* stream) {
  REQUIRE(op && ws && r && y && pair_out && row >= 0 && r != y, DSEA_ERR_ARG);
  REQUIRE(aligned16(y), DSEA_ERR_ALIGN);
  int rc = dsea_plz_correct(ws, Q, ldq, op->d.n, row, c, r, pair_out, stream);
  if (rc != DSEA_OK) return rc;
  Workspace& w = ws->w;
  int nb = launch_spmv(op->d, r, y, nullptr, nullptr, nullptr, static_cast<hipStream_t>(stream),
                       w.prof ? w.prof->next(PROF_SPMV) : nullptr);
  if (nb < 0) return DSEA_ERR_UNSUPPORTED;
  return check_launch();
}

int dsea_plz_finish(dsea_ws_t ws, const double* r, const double* y, const double* pair, double* q_out, int row,
                    double* u_out, double* alpha_out, double* beta_out, int64_t n, void* stream) {
  REQUIRE(ws && r && y && pair && q_out && u_out && alpha_out && n >= 1 && row >= 0, DSEA_ERR_ARG);
  REQUIRE(aligned16(r) && aligned16(y) && aligned16(q_out) && aligned16(u_out), DSEA_ERR_ALIGN);
  Workspace& w = ws->w;
  uint16_t* qs = nullptr;
  if (w.shadow && w.shadow_rows > row && w.shadow_ld >= n) qs = w.shadow + (int64_t)row * w.shadow_ld;
  launch_plz_finish(r, y, pair, q_out, qs, u_out, alpha_out, beta_out, n, static_cast<hipStream_t>(stream));
  return check_launch();
}

// ---------------------------------------------------------------------------- whole solvers
int dsea_lanczos_run(dsea_op_t op, dsea_ws_t ws, int k, const double* q0, double* Q, int64_t ldq,
                     double* alphas, double* betas, void* stream) {
  REQUIRE(op && ws && q0 && Q && alphas && betas && k >= 1, DSEA_ERR_ARG);
  const int64_t n = op->d.n;
  REQUIRE(ldq >= n && ws->w.n >= n, DSEA_ERR_ARG);
  REQUIRE(k <= ws->w.kmax || k == 1, DSEA_ERR_WORKSPACE);
  REQUIRE(aligned16(q0) && aligned16(Q) && (ldq % 2 == 0), DSEA_ERR_ALIGN);
  hipStream_t st = static_cast<hipStream_t>(stream);
  Workspace& w = ws->w;
  double* P = w.partials;
  double* u = w.vec[0];
  double* r = w.vec[1];
  double* nrm2 = w.scal + 0;
  TileGeom g = w.geom(n);

  // q_0 = q0/||q0|| ; u = A q_0 ; alpha_0 = q_0.u          (Lanczos.py:52-57)
  Profiler* prof = w.prof;
  // optional bf16 shadow of the basis for the correction pass (see k_axpy_norm_lp)
  uint16_t* Qs = nullptr;
  int64_t lds = 0;
  if (!w.partial_reorth && w.shadow && w.shadow_rows >= k && w.shadow_ld >= n && (!w.geom(n).split_w || n >= 32768)) {
    // small slabs (split geometry) use the split form of the shadow pass; below 2^15 rows the basis is a few MB, the
    // step is launch-bound and the 128-row fp64 split kernels expose more parallelism
    Qs = w.shadow;
    lds = w.shadow_ld;
  }
  double* lp_count = w.scal + 16;
  double* brk = w.scal + DSEA_SCAL_BREAK;  // [0] breakdown step (0 = none), [1] running max |alpha|,|beta|
  {
    hipError_t me = hipMemsetAsync(lp_count, 0, (DSEA_SCAL_BREAK + 2 - 16) * sizeof(double), st);
    if (me != hipSuccess) {
      g_last_hip = (int)me;
      return DSEA_ERR_HIP;
    }
  }
  // the lost-peer record of the single-launch form is per run (dsea_lanczos_status reads it); so is the state of the
  // partial re-orthogonalisation (scal[38..43])
  if (hipMemsetAsync(w.scal + DSEA_SCAL_LZ_FAIL, 0, (DSEA_SCAL_PRO + 4 - DSEA_SCAL_LZ_FAIL) * sizeof(double), st) != hipSuccess) {
    g_last_hip = (int)hipGetLastError();
    return DSEA_ERR_HIP;
  }
  // README-sized problems (n <= 8192, k <= 512; full-space TFIM / halo-free stencil): the whole loop as ONE launch
  // (dsea_lanczos_persist.hip).  The granule buffers live in the partial-sum area, unused by that form.
  // (automatic: up to 32 workgroups = 4096 rows, where it is measured to win; mode 1 forces it up to its envelope)
  if (w.lz_persist != 0 && w.reorth_passes == 1 && !w.partial_reorth && (w.lz_persist == 1 || n <= 4096) && !prof && lanczos_persist_applicable(op->d, n, k) &&
      lanczos_persist_comm_bytes(n, k) <= (size_t)DSEA_MAX_WAVE_TILES * (size_t)((w.kmax < 1 ? 1 : w.kmax) + 1) * sizeof(double)) {
    const int pr = launch_lanczos_persist(op->d, k, q0, Q, ldq, alphas, betas, brk, w.scal + DSEA_SCAL_LZ_FAIL, P, st, w.lose_peer);
    if (pr == -2) {
      g_last_hip = (int)hipGetLastError();
      return DSEA_ERR_HIP;
    }
    if (pr == 0) return check_launch();
  }
  // Mid-size halo-1 operators (3-point stencil, 8192 < n <= 131072 rows: BASELINE configs[2]): ONE launch that keeps a
  // third of the basis in registers and LDS and streams the rest (dsea_lanczos_persist_mid.hip).  mode 2 = off for this
  // form only (A/B measurements).
  // automatic where it is measured to win (n >= 49152 rows, k <= 400: profiles/r04_lanczos_mid_size.txt); mode 1 forces it
  // wherever it applies
  if (w.lz_persist != 0 && w.lz_persist != 2 && (w.lz_persist == 1 || (n >= 49152 && k <= 400)) && w.reorth_passes == 1 &&
      !w.partial_reorth && !prof && lanczos_persist_mid_applicable(op->d, n, k) &&
      lanczos_persist_mid_comm_bytes(n, k) <= (size_t)DSEA_MAX_WAVE_TILES * (size_t)((w.kmax < 1 ? 1 : w.kmax) + 1) * sizeof(double)) {
    const int pr = launch_lanczos_persist_mid(op->d, k, q0, Q, ldq, Qs, lds, w.lp_tau, alphas, betas, brk,
                                              w.scal + DSEA_SCAL_LZ_FAIL, lp_count, P, st, w.lose_peer);
    if (pr == -2) {
      g_last_hip = (int)hipGetLastError();
      return DSEA_ERR_HIP;
    }
    if (pr == 0) return check_launch();
  }
  const int rps = lp_rows_per_step(n, g.split_w != 0);   // 0 = split form
  const bool has_fused_tail = (op->d.kind == OP_TFIM && op->d.tfim.L_local >= 1) || (op->d.kind == OP_SELL && op->d.sell.mode == 0) ||
                              op->d.kind == OP_STENCIL3;
  // (operators without a fused tail: full re-orthogonalisation only -- refused BEFORE anything is enqueued; the partial
  //  option reaches them through dsea_lanczos_partial_step)
  if (w.partial_reorth && !has_fused_tail) return DSEA_ERR_UNSUPPORTED;
  launch_dot(q0, q0, n, P, nrm2, st);
  launch_scale_store(q0, nrm2, Q, nullptr, n, st, Qs);
  if (has_fused_tail) {
    // Fused sequence, 4 launches per step and no stand-alone scalar reductions: the mat-vec leaves
    // per-block partials of alpha (aP), the dots kernel sums them in its prologue; the axpy kernel leaves
    // per-wave partials of ||r||^2 (nP), the fused scale + mat-vec kernel sums those.
    double* aP = w.aux;
    double* nP = w.aux + DSEA_MAX_WAVE_TILES;
    int na = launch_spmv(op->d, Q, u, nullptr, nullptr, aP, st, prof ? prof->next(PROF_SPMV) : nullptr);
    if (na < 0) return DSEA_ERR_UNSUPPORTED;
    // partial re-orthogonalisation (option): the three-term vector goes to vec[2] with its norm, k_pro_update advances the
    // orthogonality estimates and decides, the dots / correction kernels then run over the basis or over nothing
    double* pro_flag = w.scal + DSEA_SCAL_PRO;
    double* pro_om = w.aux + 4 * DSEA_MAX_WAVE_TILES;            // two rows of DSEA_MAX_WAVE_TILES (k <= DSEA_MAX_KRYLOV)
    const double pro_eps1 = 64.0 * 2.220446049250313e-16;
    for (int i = 1; i < k; ++i) {
      const double* beta_prev = (i >= 2) ? betas + (i - 2) : nullptr;
      if (w.partial_reorth) {
        // (1) r = u - alpha q - beta q' and ||r||^2 -> coef[i]  (2) the estimates decide  (3) on a re-orthogonalised step:
        // c = Q^T r (the kernel's copy of r goes to vec[2], unused), r -= Q c; otherwise both kernels return at once and
        // the tail takes ||r||^2 = coef[i]
        launch_rdots(g, Q, ldq, n, i, u, nullptr, beta_prev, r, P, nullptr, st, nullptr, aP, na, alphas + (i - 1),
                     true, brk, w.zero, false);
        launch_pro_update(alphas, betas, P + (int64_t)i * g.pstride, rdots_partial_count(g, i), w.coef + i, pro_om,
                          DSEA_MAX_WAVE_TILES, pro_flag, pro_flag + 1, i, pro_eps1, w.pro_delta, brk, st);
        launch_rdots(g, Q, ldq, n, i, r, w.zero, nullptr, w.vec[2], P, w.coef, st,
                     prof ? prof->next(PROF_RDOTS) : nullptr, nullptr, 0, nullptr, false, brk, pro_flag, true);
        launch_axpy_norm(g, Q, ldq, n, i, w.coef, r, nP, nullptr, st, prof ? prof->next(PROF_AXPY) : nullptr, brk, pro_flag);
        na = launch_tfim_fused(op->d, r, nP, g.nw, Q + (int64_t)i * ldq, u, betas + (i - 1), aP, st,
                               prof ? prof->next(PROF_SPMV) : nullptr, nullptr, brk, i);
        continue;
      }
      launch_rdots(g, Q, ldq, n, i, u, nullptr, beta_prev, r, P, w.coef, st,
                   prof ? prof->next(PROF_RDOTS) : nullptr, aP, na, alphas + (i - 1), Qs != nullptr, brk);
      int nn = g.nw;
      if (Qs)
        nn = launch_axpy_norm_lp(n, rps, Q, ldq, Qs, lds, i, w.coef, w.lp_tau, r, nP, lp_count, st,
                                 prof ? prof->next(PROF_AXPY) : nullptr, brk);
      else
        launch_axpy_norm(g, Q, ldq, n, i, w.coef, r, nP, nullptr, st, prof ? prof->next(PROF_AXPY) : nullptr, brk);
      if (w.reorth_passes == 2) {
        // CGS2 option (the reference makes ONE pass, Lanczos.py:66): c' = Q^T r of the corrected r, r -= Q c'.  The dots
        // kernel rewrites r from a snapshot (alpha = 0) so that its input and output do not alias.
        if (hipMemcpyAsync(w.vec[2], r, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess) {
          g_last_hip = (int)hipGetLastError();
          return DSEA_ERR_HIP;
        }
        launch_rdots(g, Q, ldq, n, i, w.vec[2], w.zero, nullptr, r, P, w.coef, st, nullptr, nullptr, 0, nullptr,
                     Qs != nullptr, brk);
        if (Qs)
          nn = launch_axpy_norm_lp(n, rps, Q, ldq, Qs, lds, i, w.coef, w.lp_tau, r, nP, lp_count, st, nullptr, brk);
        else
          launch_axpy_norm(g, Q, ldq, n, i, w.coef, r, nP, nullptr, st, nullptr, brk);
      }
      // beta_{i-1} ~ 0 (relative to the running |alpha|, |beta| scale): the tail records step i in brk and every
      // later launch of this run returns at once (Lanczos.py:69-70 would divide by it)
      na = launch_tfim_fused(op->d, r, nP, nn, Q + (int64_t)i * ldq, u, betas + (i - 1), aP, st,
                             prof ? prof->next(PROF_SPMV) : nullptr, Qs ? Qs + (int64_t)i * lds : nullptr, brk, i);
    }
    launch_finalize_slot(aP, na, alphas + (k - 1), brk, st);
    return check_launch();
  }
  if (w.partial_reorth) return DSEA_ERR_UNSUPPORTED;   // (operators without a fused tail: full re-orthogonalisation only)
  int nb = launch_spmv(op->d, Q, u, nullptr, nullptr, P, st, prof ? prof->next(PROF_SPMV) : nullptr);
  if (nb < 0) return DSEA_ERR_UNSUPPORTED;
  launch_finalize1(P, nb, alphas, st);
  for (int i = 1; i < k; ++i) {
    const double* beta_prev = (i >= 2) ? betas + (i - 2) : nullptr;
    launch_rdots(g, Q, ldq, n, i, u, alphas + (i - 1), beta_prev, r, P, w.coef, st,
                 prof ? prof->next(PROF_RDOTS) : nullptr, nullptr, 0, nullptr, Qs != nullptr, brk);
    if (Qs) {
      double* nP = w.aux + DSEA_MAX_WAVE_TILES;
      int nn = launch_axpy_norm_lp(n, rps, Q, ldq, Qs, lds, i, w.coef, w.lp_tau, r, nP, lp_count, st,
                                   prof ? prof->next(PROF_AXPY) : nullptr, brk);
      launch_finalize_slot(nP, nn, nrm2, brk, st);
    } else {
      launch_axpy_norm(g, Q, ldq, n, i, w.coef, r, P, nullptr, st, prof ? prof->next(PROF_AXPY) : nullptr, brk);
      launch_finalize_slot(P, g.nw, nrm2, brk, st);
    }
    if (w.reorth_passes == 2) {   // CGS2 option, see the fused sequence above
      if (hipMemcpyAsync(w.vec[2], r, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess) {
        g_last_hip = (int)hipGetLastError();
        return DSEA_ERR_HIP;
      }
      launch_rdots(g, Q, ldq, n, i, w.vec[2], w.zero, nullptr, r, P, w.coef, st, nullptr, nullptr, 0, nullptr,
                   Qs != nullptr, brk);
      if (Qs) {
        double* nP2 = w.aux + DSEA_MAX_WAVE_TILES;
        int nn2 = launch_axpy_norm_lp(n, rps, Q, ldq, Qs, lds, i, w.coef, w.lp_tau, r, nP2, lp_count, st, nullptr, brk);
        launch_finalize_slot(nP2, nn2, nrm2, brk, st);
      } else {
        launch_axpy_norm(g, Q, ldq, n, i, w.coef, r, P, nullptr, st, nullptr, brk);
        launch_finalize_slot(P, g.nw, nrm2, brk, st);
      }
    }
    double* qi = Q + (int64_t)i * ldq;
    launch_scale_store(r, nrm2, qi, betas + (i - 1), n, st, Qs ? Qs + (int64_t)i * lds : nullptr, brk, i);
    nb = launch_spmv(op->d, qi, u, nullptr, brk, P, st, prof ? prof->next(PROF_SPMV) : nullptr);
    launch_finalize_slot(P, nb, alphas + i, brk, st);
  }
  return check_launch();
}

int dsea_lanczos_run_basisfree(dsea_op_t op, dsea_ws_t ws, int k, const double* q0, double* Qrot, int64_t ldq,
                               double* alphas, double* betas, const double* s, double* psi, void* stream) {
  REQUIRE(op && ws && q0 && Qrot && alphas && betas && k >= 1 && ((s == nullptr) == (psi == nullptr)), DSEA_ERR_ARG);
  const int64_t n = op->d.n;
  REQUIRE(ldq >= n && ws->w.n >= n, DSEA_ERR_ARG);
  REQUIRE(aligned16(q0) && aligned16(Qrot) && (!psi || aligned16(psi)) && (ldq % 2 == 0), DSEA_ERR_ALIGN);
  const bool has_fused_tail = (op->d.kind == OP_TFIM && op->d.tfim.L_local >= 1) || (op->d.kind == OP_SELL && op->d.sell.mode == 0) ||
                              op->d.kind == OP_STENCIL3;
  REQUIRE(has_fused_tail, DSEA_ERR_UNSUPPORTED);
  hipStream_t st = static_cast<hipStream_t>(stream);
  Workspace& w = ws->w;
  double* u = w.vec[0];
  double* r = w.vec[1];
  double* nrm2 = w.scal + 0;
  double* aP = w.aux;
  double* nP = w.aux + DSEA_MAX_WAVE_TILES;
  double* brk = w.scal + DSEA_SCAL_BREAK;
  if (hipMemsetAsync(brk, 0, 2 * sizeof(double), st) != hipSuccess ||
      hipMemsetAsync(w.scal + DSEA_SCAL_LZ_FAIL, 0, sizeof(double), st) != hipSuccess) {
    g_last_hip = (int)hipGetLastError();
    return DSEA_ERR_HIP;
  }
  if (psi && hipMemsetAsync(psi, 0, (size_t)n * sizeof(double), st) != hipSuccess) {
    g_last_hip = (int)hipGetLastError();
    return DSEA_ERR_HIP;
  }
  auto slot = [&](int i) { return Qrot + (int64_t)(i % 3) * ldq; };
  launch_dot(q0, q0, n, w.partials, nrm2, st);
  launch_scale_store(q0, nrm2, slot(0), nullptr, n, st, nullptr);
  int na = launch_spmv(op->d, slot(0), u, nullptr, nullptr, aP, st);
  if (na < 0) return DSEA_ERR_UNSUPPORTED;
  for (int i = 1; i < k; ++i) {
    // psi += s[i-1] q_{i-1} rides on the pass that reads q_{i-1} anyway
    const int nn = launch_three_term(u, slot(i - 1), i >= 2 ? slot(i - 2) : nullptr, aP, na, alphas + (i - 1),
                                     i >= 2 ? betas + (i - 2) : nullptr, r, nP, psi, s ? s + (i - 1) : nullptr, n, brk,
                                     st);
    na = launch_tfim_fused(op->d, r, nP, nn, slot(i), u, betas + (i - 1), aP, st, nullptr, nullptr, brk, i);
  }
  launch_finalize_slot(aP, na, alphas + (k - 1), brk, st);
  if (psi) launch_axpy(1.0, s + (k - 1), slot(k - 1), psi, n, st);
  return check_launch();
}

int dsea_arnoldi_extend(dsea_op_t op, dsea_ws_t ws, const double* shift, double* V, int64_t ldv, int j0, int j1,
                        double* H, int ldh, void* stream) {
  REQUIRE(op && ws && V && H && j0 >= 0 && j1 > j0 && ldh >= j1 + 1, DSEA_ERR_ARG);
  const int64_t n = op->d.n;
  REQUIRE(ldv >= n && ws->w.n >= n, DSEA_ERR_ARG);
  REQUIRE(j1 + 1 <= ws->w.kmax, DSEA_ERR_WORKSPACE);
  REQUIRE(aligned16(V) && (ldv % 2 == 0), DSEA_ERR_ALIGN);
  hipStream_t st = static_cast<hipStream_t>(stream);
  Workspace& w = ws->w;
  double* brk = w.scal + DSEA_SCAL_BREAK;
  if (j0 == 0) {   // a new factorisation: clear the break record (a continued one keeps it)
    // second-pass counter (dsea_arnoldi_second_passes) and the break record
    if (hipMemsetAsync(w.scal + 31, 0, sizeof(double), st) != hipSuccess ||
        hipMemsetAsync(w.scal + DSEA_SCAL_LZ_FAIL, 0, sizeof(double), st) != hipSuccess ||
        hipMemsetAsync(brk, 0, 2 * sizeof(double), st) != hipSuccess) {
      g_last_hip = (int)hipGetLastError();
      return DSEA_ERR_HIP;
    }
  }
  for (int j = j0; j < j1; ++j) {
    if (arnoldi_step(op->d, w, shift ? shift : w.zero, V, ldv, j, H + (int64_t)j * ldh, brk, w.scal + 24, w.scal + 26,
                     w.scal + 27, st, w.arnoldi_optimistic != 0) != 0)
      return DSEA_ERR_UNSUPPORTED;
  }
  return check_launch();
}

int dsea_ws_set_arnoldi_optimistic(dsea_ws_t ws, int on) {
  if (!ws || (on != 0 && on != 1)) return DSEA_ERR_ARG;
  ws->w.arnoldi_optimistic = on;
  return DSEA_OK;
}

int dsea_arnoldi_status(dsea_ws_t ws, int* break_step, int* redo_step, void* stream) {
  if (!ws) return DSEA_ERR_ARG;
  double h = 0.0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  double* brk = ws->w.scal + DSEA_SCAL_BREAK;
  if (hipMemcpyAsync(&h, brk, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
    g_last_hip = (int)hipGetLastError();
    return DSEA_ERR_HIP;
  }
  if (break_step) *break_step = h > 0.0 ? (int)h : 0;
  if (redo_step) *redo_step = -1;
  if (h < 0.0) {
    // optimistic mode: step -h - 1 needs its second Gram-Schmidt pass.  The record is cleared here so that the caller
    // can continue: repeat that step with the option off, then go on.
    if (redo_step) *redo_step = (int)(-h) - 1;
    if (hipMemsetAsync(brk, 0, sizeof(double), st) != hipSuccess) {
      g_last_hip = (int)hipGetLastError();
      return DSEA_ERR_HIP;
    }
    return DSEA_ERR_SECOND_PASS;
  }
  return h != 0.0 ? DSEA_ERR_BREAKDOWN : DSEA_OK;
}

int dsea_arnoldi_clear_record(dsea_ws_t ws, void* stream) {
  if (!ws) return DSEA_ERR_ARG;
  if (hipMemsetAsync(ws->w.scal + DSEA_SCAL_BREAK, 0, sizeof(double), static_cast<hipStream_t>(stream)) != hipSuccess) {
    g_last_hip = (int)hipGetLastError();
    return DSEA_ERR_HIP;
  }
  return DSEA_OK;
}
int dsea_arnoldi_status_enqueue(dsea_ws_t ws, double* host_record, void* stream) {
  if (!ws || !host_record) return DSEA_ERR_ARG;
  if (hipMemcpyAsync(host_record, ws->w.scal + DSEA_SCAL_BREAK, sizeof(double), hipMemcpyDeviceToHost,
                     static_cast<hipStream_t>(stream)) != hipSuccess) {
    g_last_hip = (int)hipGetLastError();
    return DSEA_ERR_HIP;
  }
  return DSEA_OK;
}

int dsea_arnoldi_orth(dsea_ws_t ws, const double* u, const double* shift, double* V, int64_t ldv, int64_t n, int j,
                      double* H, int ldh, void* stream) {
  REQUIRE(ws && u && V && H && j >= 0 && ldh >= j + 2 && n >= 1 && ldv >= n && ws->w.n >= n, DSEA_ERR_ARG);
  REQUIRE(j + 2 <= ws->w.kmax, DSEA_ERR_WORKSPACE);
  REQUIRE(aligned16(V) && aligned16(u) && (ldv % 2 == 0), DSEA_ERR_ALIGN);
  hipStream_t st = static_cast<hipStream_t>(stream);
  Workspace& w = ws->w;
  double* brk = w.scal + DSEA_SCAL_BREAK;
  if (j == 0 && (hipMemsetAsync(brk, 0, 2 * sizeof(double), st) != hipSuccess ||
                 hipMemsetAsync(w.scal + DSEA_SCAL_LZ_FAIL, 0, sizeof(double), st) != hipSuccess)) {
    g_last_hip = (int)hipGetLastError();
    return DSEA_ERR_HIP;
  }
  arnoldi_orth(w, n, u, shift ? shift : w.zero, V, ldv, j, H + (int64_t)j * ldh, brk, w.scal + 24, w.scal + 26,
               w.scal + 27, st);
  return check_launch();
}

int dsea_arnoldi_second_passes(dsea_ws_t ws, int64_t* count, void* stream) {
  if (!ws || !count) return DSEA_ERR_ARG;
  double h = 0.0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemcpyAsync(&h, ws->w.scal + 31, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) {
    g_last_hip = (int)hipGetLastError();
    return DSEA_ERR_HIP;
  }
  *count = (int64_t)h;
  return DSEA_OK;
}

size_t dsea_gmres_work_doubles(int m) { return m < 1 ? 0 : (size_t)(m + 1) * m + 4 * (size_t)m + 8; }

namespace {
struct GmresWork {
  double *H, *cs, *sn, *g, *y;
  int ldh;
};
inline GmresWork gmres_carve(double* work, int m) {
  GmresWork gw;
  gw.ldh = m + 1;
  gw.H = work;
  gw.cs = gw.H + (size_t)gw.ldh * m;
  gw.sn = gw.cs + m;
  gw.g = gw.sn + m;
  gw.y = gw.g + (m + 1);
  return gw;
}
}  // namespace

int dsea_gmres_begin(dsea_ws_t ws, const double* b, const double* Ax, double* V, int64_t ldv, int64_t n, int m,
                     double* work, double target, double* state, void* stream) {
  REQUIRE(ws && b && V && work && state && m >= 1 && m <= 64 && n >= 1 && ldv >= n && ws->w.n >= n, DSEA_ERR_ARG);
  REQUIRE(m + 1 <= ws->w.kmax, DSEA_ERR_WORKSPACE);
  REQUIRE(aligned16(V) && aligned16(b) && (!Ax || aligned16(Ax)) && (ldv % 2 == 0), DSEA_ERR_ALIGN);
  hipStream_t st = static_cast<hipStream_t>(stream);
  Workspace& w = ws->w;
  GmresWork gw = gmres_carve(work, m);
  double* nrm0 = w.scal + 0;
  double* r0 = w.vec[3];
  launch_residual(b, Ax, r0, n, w.partials, nrm0, st);                 // r0 = b - (A - shift) x  (Ax null: x = 0)
  launch_gmres_begin(nrm0, target, gw.g, m, state, w.scal + 28, st);
  // v0 = r0 / ||r0|| -- skipped on the device when the cycle starts converged (r0 = 0 would give 0/0)
  launch_scale_store(r0, nrm0, V, nullptr, n, st, nullptr, w.scal + 28, 0);
  return check_launch();
}

int dsea_gmres_step(dsea_op_t op, dsea_ws_t ws, const double* shift, const double* u, double* V, int64_t ldv,
                    int64_t n, int j, int m, double* work, double target, double* state, void* stream) {
  REQUIRE(ws && V && work && state && m >= 1 && m <= 64 && j >= 0 && j < m && (op || u), DSEA_ERR_ARG);
  if (op) n = op->d.n;
  REQUIRE(n >= 1 && ldv >= n && ws->w.n >= n, DSEA_ERR_ARG);
  hipStream_t st = static_cast<hipStream_t>(stream);
  Workspace& w = ws->w;
  GmresWork gw = gmres_carve(work, m);
  double* brk = w.scal + 28;          // skip / breakdown record of this cycle's Arnoldi steps
  const double* sh = shift ? shift : w.zero;
  if (op) {
    if (arnoldi_step(op->d, w, sh, V, ldv, j, gw.H + (size_t)j * gw.ldh, brk, w.scal + 24, w.scal + 26, w.scal + 27,
                     st, w.arnoldi_optimistic != 0) != 0)
      return DSEA_ERR_UNSUPPORTED;
  } else {
    REQUIRE(aligned16(u), DSEA_ERR_ALIGN);
    arnoldi_orth(w, n, u, sh, V, ldv, j, gw.H + (size_t)j * gw.ldh, brk, w.scal + 24, w.scal + 26, w.scal + 27, st);
  }
  launch_gmres_givens(gw.H, gw.ldh, j, gw.cs, gw.sn, gw.g, target, state, brk, st);
  return check_launch();
}

int dsea_gmres_end(dsea_ws_t ws, const double* V, int64_t ldv, int64_t n, int m, double* work, const double* state,
                   double* x, void* stream) {
  REQUIRE(ws && V && work && state && x && m >= 1 && m <= 64 && n >= 1 && ldv >= n && ws->w.n >= n, DSEA_ERR_ARG);
  REQUIRE(aligned16(V) && aligned16(x) && (ldv % 2 == 0), DSEA_ERR_ALIGN);
  hipStream_t st = static_cast<hipStream_t>(stream);
  Workspace& w = ws->w;
  GmresWork gw = gmres_carve(work, m);
  launch_gmres_solve(gw.H, gw.ldh, m, gw.g, state, gw.y, st);
  // x += V[0..m) y   (y is zero beyond the columns processed; V must hold finite values there: callers zero it once)
  TileGeom tg = w.geom(n);
  double* dx = w.vec[0];
  launch_ritz(tg, V, ldv, n, m, gw.y, dx, st);
  launch_axpy(1.0, nullptr, dx, x, n, st);
  return check_launch();
}

int dsea_gmres_cycle(dsea_op_t op, dsea_ws_t ws, const double* shift, const double* b, double* x, double* V,
                     int64_t ldv, int m, double* work, double target, double* state, int first, void* stream) {
  REQUIRE(op && ws && b && x && V && work && state && m >= 1 && m <= 64, DSEA_ERR_ARG);
  const int64_t n = op->d.n;
  REQUIRE(aligned16(x), DSEA_ERR_ALIGN);
  hipStream_t st = static_cast<hipStream_t>(stream);
  Workspace& w = ws->w;
  const double* Ax = nullptr;
  if (!first) {   // r0 = b - (A - shift) x
    double* u = w.vec[0];
    int nb = launch_spmv(op->d, x, u, shift, nullptr, w.partials, st);
    if (nb < 0) return DSEA_ERR_UNSUPPORTED;
    Ax = u;
  }
  int rc = dsea_gmres_begin(ws, b, Ax, V, ldv, n, m, work, target, state, stream);
  if (rc != DSEA_OK) return rc;
  for (int j = 0; j < m; ++j) {
    rc = dsea_gmres_step(op, ws, shift, nullptr, V, ldv, n, j, m, work, target, state, stream);
    if (rc != DSEA_OK) return rc;
  }
  return dsea_gmres_end(ws, V, ldv, n, m, work, state, x, stream);
}

int dsea_lanczos_status(dsea_ws_t ws, int* break_step, void* stream) {
  if (!ws) return DSEA_ERR_ARG;
  double h[DSEA_SCAL_LZ_FAIL - DSEA_SCAL_BREAK + 1];
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemcpyAsync(h, ws->w.scal + DSEA_SCAL_BREAK, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) {
    g_last_hip = (int)hipGetLastError();
    return DSEA_ERR_HIP;
  }
  if (break_step) *break_step = (int)h[0];
  if (h[DSEA_SCAL_LZ_FAIL - DSEA_SCAL_BREAK] != 0.0) return DSEA_ERR_TIMEOUT;   // single-launch form: a peer was lost
  return h[0] != 0.0 ? DSEA_ERR_BREAKDOWN : DSEA_OK;
}

int dsea_cg_run(dsea_op_t op, dsea_ws_t ws, const double* shift, const double* b, double* x, double* state,
                double eps, int64_t maxiter, int poll_every, int64_t* iters_out, double* resnorm_out,
                void* stream) {
  REQUIRE(op && ws && b && x && state && maxiter >= 0, DSEA_ERR_ARG);
  const int64_t n = op->d.n;
  REQUIRE(ws->w.n >= n, DSEA_ERR_ARG);
  REQUIRE(aligned16(b) && aligned16(x), DSEA_ERR_ALIGN);
  hipStream_t st = static_cast<hipStream_t>(stream);
  Workspace& w = ws->w;
  double* P = w.partials;
  double* r = w.vec[1];
  double* d = w.vec[2];
  double* Ad = w.vec[3];
  const double* done = state + DSEA_CG_DONE;
  double* dP = w.aux;                            // partials of d.A'd (one per mat-vec block)
  double* rP = w.aux + DSEA_MAX_WAVE_TILES;      // partials of r.r   (one per update block)
  if (poll_every <= 0) poll_every = 16;

  // Small halo-1 operators: the whole solve in ONE persistent launch (k_cg_persist_stencil), bit-identical iterates.
  // Its granule buffer lives in w.aux (the streaming form's partial buffers, unused there).  README-sized full-space
  // TFIM operators (n <= 8192): k_cg_persist_tfim, granules in the partial-sum area.
  const bool tfim_persist = w.persist_override != 0 && cg_persist_tfim_applicable(op->d) &&
                            cg_persist_tfim_comm_bytes(n) <=
                                (size_t)DSEA_MAX_WAVE_TILES * (size_t)((w.kmax < 1 ? 1 : w.kmax) + 1) * sizeof(double);
  // 2^11 ... 2^20 rows: k_cg_persist_tfim_big (iterates bit-identical to the streaming form; d double-buffered in the
  // workspace vectors the streaming form uses for d and A'd, granules in w.aux)
  const bool tfim_big = w.persist_override != 0 && cg_persist_tfim_big_applicable(op->d) &&
                        cg_persist_tfim_big_comm_bytes(n) <= (size_t)4 * DSEA_MAX_WAVE_TILES * sizeof(double);
  if (tfim_big || tfim_persist ||
      (w.persist_override != 0 && persist_comm_bytes(n) <= (size_t)4 * DSEA_MAX_WAVE_TILES * sizeof(double))) {
    // (persist_override 200 = the two-exchange form whose iterates are bit-identical to the streaming kernels; default: the
    //  one-exchange form, csrc/dsea_cg_persist_tfim_big.hip MERGED)
    const int pr = tfim_big ? launch_cg_persist_tfim_big(op->d, shift, b, x, state, eps, maxiter, w.aux, d, Ad, st, w.lose_peer,
                                                         w.persist_override != 200)
                   : tfim_persist ? launch_cg_persist_tfim(op->d, shift, b, x, state, eps, maxiter, P, st, w.lose_peer)
                                  : launch_cg_persist(op->d, shift, b, x, state, eps, maxiter, w.aux,
                                                      (w.persist_override > 0 && w.persist_override != 200) ? w.persist_override : 0, st, w.lose_peer);
    if (pr == -2) {
      g_last_hip = (int)hipGetLastError();
      return DSEA_ERR_HIP;
    }
    if (pr == 0) {
      const bool merged_form = tfim_big ? (w.persist_override != 200)
                                        : (!tfim_persist && w.persist_override >= 100 && w.persist_override != 200);
      w.last_cg_form = merged_form ? DSEA_CG_FORM_PERSISTENT_MERGED : DSEA_CG_FORM_PERSISTENT;
      double hs[DSEA_CG_STATE_LEN];
      if (hipMemcpyAsync(hs, state, sizeof(hs), hipMemcpyDeviceToHost, st) != hipSuccess ||
          hipStreamSynchronize(st) != hipSuccess) {
        g_last_hip = (int)hipGetLastError();
        return DSEA_ERR_HIP;
      }
      if (iters_out) *iters_out = (int64_t)hs[DSEA_CG_ITERS];
      if (resnorm_out) *resnorm_out = hs[DSEA_CG_RESNORM];
      int rc0 = check_launch();
      if (rc0 != DSEA_OK) return rc0;
      if (hs[DSEA_CG_DONE] < 0.0) return DSEA_ERR_TIMEOUT;   // a workgroup of the persistent launch did not show up
      return hs[DSEA_CG_DONE] != 0.0 ? DSEA_OK : DSEA_ERR_NOT_CONVERGED;
    }
  }

  w.last_cg_form = DSEA_CG_FORM_STREAMING;
  // r = b - A'x0 ; early out ; d = r                            (CG.py:26-30)
  int nb = launch_spmv(op->d, x, Ad, shift, nullptr, nullptr, st);
  if (nb < 0) return DSEA_ERR_UNSUPPORTED;
  launch_cg_init(b, Ad, r, d, state, n, P, st);
  launch_cg_init_check(state, eps, st);

  double host_state[DSEA_CG_STATE_LEN];
  int64_t issued = 0;
  auto issue_chunk = [&]() -> int {      // the next <= poll_every iterations; returns how many were enqueued
    const int64_t chunk = (maxiter - issued) < poll_every ? (maxiter - issued) : poll_every;
    for (int64_t it = 0; it < chunk; ++it) {
      const int parity = (int)((issued + it) & 1);
      nb = launch_spmv(op->d, d, Ad, shift, done, dP, st);                              // A'd, d.A'd partials (CG.py:31/40)
      const int nr = launch_cg_update_fused(x, r, d, Ad, state, parity, dP, nb, n, rP, st);  // CG.py:31,33-34
      launch_cg_direction_fused(r, d, state, parity, rP, nr, eps, n, st);              // CG.py:35-39
    }
    issued += chunk;
    return (int)chunk;
  };
  StatePoller* sp = state_poller();
  if (sp) {
    // Pipelined polling: chunk j + 1 is enqueued BEFORE the host looks at the state left by chunk j, so the device never
    // idles across a host round trip (measured: ~55 us per poll at the headline size).  Launches issued after
    // convergence are no-ops on the device (DONE flag), at most one chunk of them.
    int slot = 0;
    auto snapshot = [&](int sl) -> bool {
      return hipMemcpyAsync(sp->pinned + sl * DSEA_CG_STATE_LEN, state, sizeof(host_state), hipMemcpyDeviceToHost, st) ==
                 hipSuccess &&
             hipEventRecord(sp->ev[sl], st) == hipSuccess;
    };
    bool okh = snapshot(slot);                 // the state after the initial residual (early out, CG.py:28-29)
    while (okh) {
      const bool more = issued < maxiter;
      if (more) {
        issue_chunk();
        okh = snapshot(slot ^ 1);
        if (!okh) break;
      }
      if (hipEventSynchronize(sp->ev[slot]) != hipSuccess) {
        okh = false;
        break;
      }
      memcpy(host_state, sp->pinned + slot * DSEA_CG_STATE_LEN, sizeof(host_state));
      if (host_state[DSEA_CG_DONE] != 0.0 || !more) {
        if (more) {   // one chunk was enqueued behind the converged state: wait for it, its snapshot is the final state
          if (hipEventSynchronize(sp->ev[slot ^ 1]) != hipSuccess) {
            okh = false;
            break;
          }
          memcpy(host_state, sp->pinned + (slot ^ 1) * DSEA_CG_STATE_LEN, sizeof(host_state));
        }
        break;
      }
      slot ^= 1;
    }
    if (!okh) {
      g_last_hip = (int)hipGetLastError();
      return DSEA_ERR_HIP;
    }
  } else {
    bool finished = false;
    while (!finished) {
      issue_chunk();
      if (hipMemcpyAsync(host_state, state, sizeof(host_state), hipMemcpyDeviceToHost, st) != hipSuccess ||
          hipStreamSynchronize(st) != hipSuccess) {
        g_last_hip = (int)hipGetLastError();
        return DSEA_ERR_HIP;
      }
      finished = (host_state[DSEA_CG_DONE] != 0.0) || issued >= maxiter;
    }
  }
  if (iters_out) *iters_out = (int64_t)host_state[DSEA_CG_ITERS];
  if (resnorm_out) *resnorm_out = host_state[DSEA_CG_RESNORM];
  int rc = check_launch();
  if (rc != DSEA_OK) return rc;
  return host_state[DSEA_CG_DONE] != 0.0 ? DSEA_OK : DSEA_ERR_NOT_CONVERGED;
}

}  // extern "C"
