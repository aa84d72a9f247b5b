// C ABI of the fp64 device path (declared in include/rsparse_wrmf_hip.h, section 3) and the two stateless `*_double`
// drop-ins, which run through it: als_implicit_double / als_explicit_double (src/wrmf_implicit.cpp:5-14,
// src/wrmf_explicit.cpp:5-14) compute in double in the reference, and so do these.  Kernels: wrmf_f64.hip.
#include "../../include/rsparse_wrmf_hip.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <string>
#include <vector>

#include "wrmf_f64.h"
#include "wrmf_internal.h"

using namespace rsparse_hip;

struct rsparse_hip_csc_f64 {
  int n_rows = 0, n_cols = 0;
  int64_t nnz = 0;
  const int32_t* col_ptrs = nullptr;
  const int32_t* row_idx = nullptr;
  double* vals = nullptr;
  bool owns = false;
  int device = 0;
  // the rows the conjugate-gradient path cuts into chunks (wrmf_f64.hip, "long rows"): listed once, from the column pointers
  int long_min = 0x7fffffff, chunk_len = 0, n_long = 0, n_chunks = 0;
  int32_t* long_table = nullptr;   // device: long_rows [n_long] | long_chunk0 [n_long + 1] | chunk_long [n_chunks] | chunk_off [n_chunks]
  ~rsparse_hip_csc_f64() { if (long_table) (void)hipFree(long_table); }
};

namespace {

#define HIP_TRY(expr)                                       \
  do {                                                      \
    hipError_t _e = (expr);                                 \
    if (_e != hipSuccess) return capi_hip_fail(_e, #expr);  \
  } while (0)

int fail(int code, const std::string& msg) { return capi_fail(code, msg); }

constexpr size_t kRhsInitDoubles = (size_t)256 * 128 + 128;

// Rows of more than g_long_row_min non-zeros leave the wave-per-row kernel (a 2048-row is 128 batches x 5 passes ~ 1 ms of one wave; the
// longest row of the 1M x 100k timing matrix, 154 k, was 58-80 ms); rsparse_hip_set_f64_long_rows changes the two lengths for the
// handles made afterwards (tests: small matrices)
constexpr int kF64LongRowMin = 2048, kF64LongRowChunk = 1024;
int g_long_row_min = kF64LongRowMin, g_long_row_chunk = kF64LongRowChunk;

int list_long_rows(rsparse_hip_csc_f64& m, const int32_t* host_col_ptrs) {
  const int lmin = g_long_row_min, chunk = g_long_row_chunk;
  m.long_min = lmin; m.chunk_len = chunk; m.n_long = m.n_chunks = 0;
  std::vector<int32_t> rows, chunk0, cl, co;
  for (int c = 0; c < m.n_cols; c++) {
    const int n = host_col_ptrs[(size_t)c + 1] - host_col_ptrs[(size_t)c];
    if (n <= lmin) continue;
    chunk0.push_back((int32_t)cl.size());
    for (int off = 0; off < n; off += chunk) {
      cl.push_back((int32_t)rows.size());
      co.push_back(off);
    }
    rows.push_back(c);
  }
  if (rows.empty()) return RSPARSE_HIP_OK;
  chunk0.push_back((int32_t)cl.size());
  std::vector<int32_t> all;
  all.reserve(rows.size() + chunk0.size() + 2 * cl.size());
  all.insert(all.end(), rows.begin(), rows.end());
  all.insert(all.end(), chunk0.begin(), chunk0.end());
  all.insert(all.end(), cl.begin(), cl.end());
  all.insert(all.end(), co.begin(), co.end());
  HIP_TRY(hipMalloc(&m.long_table, all.size() * sizeof(int32_t)));
  HIP_TRY(hipMemcpy(m.long_table, all.data(), all.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  m.n_long = (int)rows.size();
  m.n_chunks = (int)cl.size();
  return RSPARSE_HIP_OK;
}

// grow-only scratch of the fp64 path (one host thread drives the library, as in wrmf_capi.cpp)
struct WorkspaceF64 {
  double* gram = nullptr;
  size_t gram_n = 0;
  double* partials = nullptr;   // per-workgroup loss terms + the tail of the two-stage sum
  double* scalars = nullptr;    // [0] loss rows, [1] sumsq, [2], [3] sums of the bias sweeps
  double* rinit = nullptr;      // 256 x 128 partials + the k1 entries of rhs_init
  double* m2 = nullptr;
  size_t m2_n = 0;
  double* longs = nullptr;      // the long rows' partial sums and vectors (f64_long_scratch_doubles)
  size_t longs_n = 0;
  double* repack = nullptr;     // explicit feedback with biases, conjugate gradient: X', Y', shifted ratings
  size_t repack_n = 0;
  int device = -1;
  void release() {
    for (double** q : {&gram, &partials, &scalars, &rinit, &m2, &longs, &repack})
      if (*q) { (void)hipFree(*q); *q = nullptr; }
    gram_n = m2_n = longs_n = repack_n = 0;
  }
  int ensure_repack(size_t n) {
    if (n > repack_n) {
      if (repack) (void)hipFree(repack);
      repack = nullptr; repack_n = 0;
      HIP_TRY(hipMalloc(&repack, n * sizeof(double)));
      repack_n = n;
    }
    return RSPARSE_HIP_OK;
  }
  int ensure_longs(size_t n) {
    if (n > longs_n) {
      if (longs) (void)hipFree(longs);
      longs = nullptr; longs_n = 0;
      HIP_TRY(hipMalloc(&longs, n * sizeof(double)));
      longs_n = n;
    }
    return RSPARSE_HIP_OK;
  }
  int ensure() {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != device) {
      release();
      device = dev;
    }
    if (!partials) HIP_TRY(hipMalloc(&partials, ((size_t)std::max(kF64MaxGrid, 1024) + kSumStageBlocks + 16) * sizeof(double)));
    if (!scalars) {
      HIP_TRY(hipMalloc(&scalars, 16 * sizeof(double)));
      HIP_TRY(hipMemset(scalars, 0, 16 * sizeof(double)));
    }
    if (!rinit) HIP_TRY(hipMalloc(&rinit, kRhsInitDoubles * sizeof(double)));
    return RSPARSE_HIP_OK;
  }
  int ensure_gram(size_t n) {
    if (n > gram_n) {
      if (gram) (void)hipFree(gram);
      gram = nullptr; gram_n = 0;
      HIP_TRY(hipMalloc(&gram, n * sizeof(double)));
      gram_n = n;
    }
    return RSPARSE_HIP_OK;
  }
  int ensure_m2(size_t n) {
    if (n > m2_n) {
      if (m2) (void)hipFree(m2);
      m2 = nullptr; m2_n = 0;
      HIP_TRY(hipMalloc(&m2, n * sizeof(double)));
      m2_n = n;
    }
    return RSPARSE_HIP_OK;
  }
} g_w64;

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
  template <class T> T* as() { return static_cast<T*>(p); }
};

// One half-iteration in double.  d_base_in / d_base_out: global_bias_base of the no-bias global-bias variant (`rank`
// doubles; in = use it instead of recomputing -global_bias * rowSums(X), out = receives what was used), both nullable.
int f64_half_iteration(const rsparse_hip_csc_f64* conf, bool implicit, const double* d_X, double* d_Y, const double* d_XtX,
                       int rank, double lambda, unsigned solver, unsigned cg_steps, int dynamic_lambda, int with_biases,
                       int is_x_bias_last_row, double global_bias, const double* d_base_in, double* d_base_out,
                       double* d_loss_rows_out, hipStream_t s) {
  if (!conf) return fail(RSPARSE_HIP_ERR_INVALID, "conf is NULL");
  if (!d_X || !d_Y) return fail(RSPARSE_HIP_ERR_INVALID, "X or Y is NULL");
  if (implicit && !d_XtX) return fail(RSPARSE_HIP_ERR_INVALID, "XtX is NULL");
  if (rank <= 0) return fail(RSPARSE_HIP_ERR_INVALID, "rank must be positive");
  if (rank > RSPARSE_HIP_MAX_RANK_F64) return fail(RSPARSE_HIP_ERR_UNSUPPORTED, "rank > 128 is not on the fp64 device path");
  if (solver > RSPARSE_SOLVER_NNLS) return fail(RSPARSE_HIP_ERR_INVALID, "unknown solver code");
  if (with_biases && rank < 2) return fail(RSPARSE_HIP_ERR_INVALID, "with_biases needs rank >= 2 (a row of ones and a bias row)");
  if (with_biases && implicit && solver == RSPARSE_SOLVER_CONJUGATE_GRADIENT)
    // the reference drops a row of the warm start twice on this path (wrmf_implicit.hpp:189,197) and cannot run it
    return fail(RSPARSE_HIP_ERR_UNSUPPORTED, "with_user_item_bias + conjugate_gradient with implicit feedback is not on the device path");
  int rc = g_w64.ensure();
  if (rc) return rc;
  int* fails = capi_fail_counters();
  if (!fails) return RSPARSE_HIP_ERR_RUNTIME;
  double* out = d_loss_rows_out ? d_loss_rows_out : g_w64.scalars;
  if (conf->n_cols == 0) {
    HIP_TRY(hipMemsetAsync(out, 0, sizeof(double), s));
    return RSPARSE_HIP_OK;
  }
  // wrmf_implicit.hpp:108-109: a global bias below sqrt(eps) of the element type counts as zero (double: 1.49e-8)
  const double gb = (implicit && global_bias >= std::sqrt(DBL_EPSILON)) ? global_bias : 0.0;
  F64Args a;
  a.col_ptrs = conf->col_ptrs; a.row_idx = conf->row_idx; a.vals = conf->vals;
  a.X = d_X; a.Y = d_Y; a.XtX = implicit ? d_XtX : nullptr;
  a.n_cols = conf->n_cols; a.k = rank; a.k1 = with_biases ? rank - 1 : rank;
  a.xoff = with_biases ? (is_x_bias_last_row ? 0 : 1) : 0;            // first kept row of X_nnz         (:88 / :188)
  a.xb = with_biases ? (is_x_bias_last_row ? rank - 1 : 0) : -1;      // row of X holding the x biases   (:59-64 / :116-120)
  a.ioff = with_biases ? (is_x_bias_last_row ? 1 : 0) : 0;            // drop_row(init, !is_x_bias_last_row)
  a.ooff = with_biases ? (is_x_bias_last_row ? 0 : 1) : 0;            // head / tail of Y.col(i)
  a.implicit = implicit ? 1 : 0; a.solver = (int)solver; a.cg_steps = (int)cg_steps;
  a.lambda = lambda; a.dynamic_lambda = dynamic_lambda ? 1 : 0;
  a.gbias = gb;
  a.rhs_init = nullptr;
  a.solve_empty = (implicit && (with_biases || gb != 0.0)) ? 1 : 0;   // wrmf_implicit.hpp:178
  double* rinit = g_w64.rinit + (kRhsInitDoubles - 128);
  hipError_t e;
  if (implicit && with_biases) {        // rhs_init = -X' (x_b + global_bias)  (:142-153)
    if ((e = launch_f64_rhs_init(d_X, rank, a.xoff, a.k1, a.xb, gb, conf->n_rows, g_w64.rinit, rinit, s)) != hipSuccess)
      return capi_hip_fail(e, "launch_f64_rhs_init");
    a.rhs_init = rinit;
  } else if (gb != 0.0) {               // global_bias_base = -global_bias * rowSums(X)  (:110-112, :155-157)
    if (d_base_in) {
      HIP_TRY(hipMemcpyAsync(rinit, d_base_in, (size_t)rank * sizeof(double), hipMemcpyDeviceToDevice, s));
    } else if ((e = launch_f64_rhs_init(d_X, rank, 0, rank, -1, gb, conf->n_rows, g_w64.rinit, rinit, s)) != hipSuccess) {
      return capi_hip_fail(e, "launch_f64_rhs_init");
    }
    if (d_base_out) HIP_TRY(hipMemcpyAsync(d_base_out, rinit, (size_t)rank * sizeof(double), hipMemcpyDeviceToDevice, s));
    a.rhs_init = rinit;
  }
  const int grid = f64_als_grid(conf->n_cols);
  a.loss_partials = g_w64.partials;
  a.fail_counter = fails;
  a.m2_scratch = nullptr;
  if (f64_needs_m2_scratch(a.k1, a.solver)) {
    if ((rc = g_w64.ensure_m2((size_t)grid * f64_m2_doubles_per_wg(a.k1)))) return rc;
    a.m2_scratch = g_w64.m2;
  }
  a.long_min = 0x7fffffff; a.chunk_len = 0; a.n_long = a.n_chunks = 0;
  a.long_rows = a.long_chunk0 = a.chunk_long = a.chunk_off = nullptr;
  a.long_scratch = nullptr;
  if (conf->n_long > 0 && solver == RSPARSE_SOLVER_CONJUGATE_GRADIENT) {   // (only the wave-per-row path looks at these)
    if ((rc = g_w64.ensure_longs(f64_long_scratch_doubles(rank, conf->n_long, conf->n_chunks)))) return rc;
    a.long_min = conf->long_min; a.chunk_len = conf->chunk_len; a.n_long = conf->n_long; a.n_chunks = conf->n_chunks;
    a.long_rows = conf->long_table;
    a.long_chunk0 = a.long_rows + conf->n_long;
    a.chunk_long = a.long_chunk0 + conf->n_long + 1;
    a.chunk_off = a.chunk_long + conf->n_chunks;
    a.long_scratch = g_w64.longs;
  }
  // explicit feedback with user/item biases and conjugate gradient -- the usual configuration of an explicit fit in the reference's
  // default precision -- re-packed for the wave-per-row kernels (end of round 6: it ran the generic kernel, 143 ms per iteration
  // at 1M x 100k and rank 10 where the float fit takes 10)
  const bool repack = !implicit && with_biases && solver == RSPARSE_SOLVER_CONJUGATE_GRADIENT && a.k1 >= 1;
  if (repack) {
    const int k1 = a.k1;
    const size_t nx = (size_t)conf->n_rows * k1, ny = (size_t)conf->n_cols * k1, nv = (size_t)conf->nnz;
    if ((rc = g_w64.ensure_repack(nx + ny + nv + 16))) return rc;
    double* Xp = g_w64.repack;
    double* Yp = Xp + nx;
    double* Vp = Yp + ny;
    if ((e = launch_f64_pack_rows(d_X, rank, a.xoff, k1, conf->n_rows, Xp, s)) != hipSuccess) return capi_hip_fail(e, "launch_f64_pack_rows");
    if ((e = launch_f64_pack_rows(d_Y, rank, a.ioff, k1, conf->n_cols, Yp, s)) != hipSuccess) return capi_hip_fail(e, "launch_f64_pack_rows");
    if ((e = launch_f64_shift_values(conf->vals, conf->row_idx, d_X, rank, a.xb, conf->nnz, Vp, s)) != hipSuccess)
      return capi_hip_fail(e, "launch_f64_shift_values");
    F64Args b = a;
    b.X = Xp; b.Y = Yp; b.vals = Vp;
    b.k = k1; b.k1 = k1;
    b.xoff = 0; b.xb = -1; b.ioff = 0; b.ooff = 0;
    if (conf->n_long > 0) {   // (the long rows' scratch was sized for `rank` coordinates: enough for k1)
      b.long_scratch = g_w64.longs;
    }
    if ((e = launch_f64_als(b, s)) != hipSuccess) return capi_hip_fail(e, "launch_f64_als");
    if ((e = launch_f64_unpack_rows(Yp, k1, conf->n_cols, rank, a.ooff, d_Y, s)) != hipSuccess) return capi_hip_fail(e, "launch_f64_unpack_rows");
  } else {
    if ((e = launch_f64_als(a, s)) != hipSuccess) return capi_hip_fail(e, "launch_f64_als");
  }
  if ((e = launch_sum_partials(g_w64.partials, (size_t)grid, out, s)) != hipSuccess) return capi_hip_fail(e, "launch_sum_partials");
  return RSPARSE_HIP_OK;
}

// Shared body of the two stateless `*_double` drop-ins.
int stateless_double(bool implicit, int n_rows, int n_cols, const int32_t* col_ptrs, const int32_t* row_indices,
                     const double* values, const double* X, double* Y, const double* XtX, const double* cnt_X, int rank,
                     double lambda, unsigned solver, unsigned cg_steps, int dynamic_lambda, double* loss_out,
                     int with_biases, int is_x_bias_last_row, double global_bias, double* global_bias_base,
                     int global_bias_base_len, int initialize_bias_base) {
  if (n_rows < 0 || n_cols < 0) return fail(RSPARSE_HIP_ERR_INVALID, "negative matrix dimension");
  if (!col_ptrs) return fail(RSPARSE_HIP_ERR_INVALID, "col_ptrs is NULL");
  if (!X || !Y) return fail(RSPARSE_HIP_ERR_INVALID, "X or Y is NULL");
  if (rank <= 0) return fail(RSPARSE_HIP_ERR_INVALID, "rank must be positive");
  if (rank > RSPARSE_HIP_MAX_RANK_F64) return fail(RSPARSE_HIP_ERR_UNSUPPORTED, "rank > 128 is not on the fp64 device path");
  if (solver > RSPARSE_SOLVER_NNLS) return fail(RSPARSE_HIP_ERR_INVALID, "unknown solver code");
  if (with_biases && implicit && solver == RSPARSE_SOLVER_CONJUGATE_GRADIENT)
    return fail(RSPARSE_HIP_ERR_UNSUPPORTED, "with_user_item_bias + conjugate_gradient with implicit feedback is not on the device path");
  if (implicit && !XtX) return fail(RSPARSE_HIP_ERR_INVALID, "XtX is NULL");
  const int64_t nnz = (int64_t)col_ptrs[n_cols] - col_ptrs[0];
  if (col_ptrs[0] != 0 || nnz < 0) return fail(RSPARSE_HIP_ERR_INVALID, "col_ptrs must start at 0 and be non-decreasing");
  for (int c = 0; c < n_cols; c++)
    if (col_ptrs[c + 1] < col_ptrs[c]) return fail(RSPARSE_HIP_ERR_INVALID, "col_ptrs is not non-decreasing");
  if (nnz > 0 && (!row_indices || !values)) return fail(RSPARSE_HIP_ERR_INVALID, "row_indices or values is NULL");
  for (int64_t e = 0; e < nnz; e++)
    if (row_indices[e] < 0 || row_indices[e] >= n_rows) return fail(RSPARSE_HIP_ERR_INVALID, "row index out of range");
  const size_t nx = (size_t)rank * n_rows, ny = (size_t)rank * n_cols;
  const size_t ng = implicit && with_biases ? (size_t)(rank - 1) * (rank - 1) : (size_t)rank * rank;
  DevBuf dP, dI, dV, dX, dY, dG, dW, dBase;
  HIP_TRY(dP.alloc(((size_t)n_cols + 1) * 4));
  HIP_TRY(dI.alloc((size_t)nnz * 4));
  HIP_TRY(dV.alloc((size_t)nnz * 8));
  HIP_TRY(dX.alloc(nx * 8));
  HIP_TRY(dY.alloc(ny * 8));
  HIP_TRY(hipMemcpy(dP.p, col_ptrs, ((size_t)n_cols + 1) * 4, hipMemcpyHostToDevice));
  if (nnz) {
    HIP_TRY(hipMemcpy(dI.p, row_indices, (size_t)nnz * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dV.p, values, (size_t)nnz * 8, hipMemcpyHostToDevice));
  }
  if (nx) HIP_TRY(hipMemcpy(dX.p, X, nx * 8, hipMemcpyHostToDevice));
  if (ny) HIP_TRY(hipMemcpy(dY.p, Y, ny * 8, hipMemcpyHostToDevice));
  if (implicit) {
    HIP_TRY(dG.alloc(ng * 8));
    HIP_TRY(hipMemcpy(dG.p, XtX, ng * 8, hipMemcpyHostToDevice));
  }
  const bool weighted = !implicit && dynamic_lambda;
  if (weighted && lambda > 0) {
    if (!cnt_X) return fail(RSPARSE_HIP_ERR_INVALID, "cnt_X is NULL with dynamic_lambda");
    HIP_TRY(dW.alloc((size_t)n_rows * 8));
    if (n_rows) HIP_TRY(hipMemcpy(dW.p, cnt_X, (size_t)n_rows * 8, hipMemcpyHostToDevice));
  }
  int rc = g_w64.ensure();
  if (rc) return rc;
  StaleFailures stale_guard;   // counters left by earlier device-resident calls are not this call's (handed back at the end)
  rsparse_hip_csc_f64 conf;
  conf.n_rows = n_rows; conf.n_cols = n_cols; conf.nnz = nnz;
  conf.col_ptrs = dP.as<int32_t>(); conf.row_idx = dI.as<int32_t>(); conf.vals = dV.as<double>();
  if ((rc = list_long_rows(conf, col_ptrs))) return rc;
  const bool gbias = implicit && !with_biases && global_bias >= std::sqrt(DBL_EPSILON);
  const double* base_in = nullptr;
  double* base_out = nullptr;
  const int blen = global_bias_base ? std::max(global_bias_base_len, 0) : 0;
  bool given = false;
  if (gbias) {
    // global_bias_base: `rank` entries (wrmf_implicit.hpp:111-112); the caller's buffer holds global_bias_base_len of them
    // (the R driver allocates rank - 1, R/model_WRMF.R:292).  Never more than the stated length is touched: it is READ
    // (initialize_bias_base == 0) only when it holds the whole vector, otherwise recomputed from X (its definition); it is
    // WRITTEN up to min(len, rank) entries
    HIP_TRY(dBase.alloc((size_t)rank * 8));
    given = !initialize_bias_base && blen >= rank;
    if (given) {
      HIP_TRY(hipMemcpy(dBase.p, global_bias_base, (size_t)rank * 8, hipMemcpyHostToDevice));
      base_in = dBase.as<double>();
    } else {
      base_out = dBase.as<double>();
    }
  }
  rc = f64_half_iteration(&conf, implicit, dX.as<double>(), dY.as<double>(), dG.as<double>(), rank, lambda, solver, cg_steps,
                          dynamic_lambda, with_biases, is_x_bias_last_row, global_bias, base_in, base_out, g_w64.scalars,
                          nullptr);
  if (rc) return rc;
  if (gbias && !given && initialize_bias_base && blen > 0) {
    std::vector<double> hb((size_t)rank);
    HIP_TRY(hipMemcpy(hb.data(), dBase.p, (size_t)rank * 8, hipMemcpyDeviceToHost));
    for (int t = 0; t < std::min(blen, rank); t++) global_bias_base[t] = hb[(size_t)t];
  }
  if (lambda > 0 && nx > 0) {  // + lambda * accu(X % X)  [* cnt_X]; with biases every row of X but the ones (:147-159, :287-297)
    const double* Xreg = dX.as<double>();
    int kreg = rank;
    DevBuf dXe;
    if (with_biases) {
      kreg = rank - 1;
      HIP_TRY(dXe.alloc((size_t)kreg * n_rows * 8));
      HIP_TRY(hipMemcpy2D(dXe.p, (size_t)kreg * 8, dX.as<double>() + (is_x_bias_last_row ? 1 : 0), (size_t)rank * 8,
                          (size_t)kreg * 8, (size_t)n_rows, hipMemcpyDeviceToDevice));
      Xreg = dXe.as<double>();
    }
    hipError_t e = launch_f64_weighted_sumsq(Xreg, kreg, n_rows, (weighted && lambda > 0) ? dW.as<double>() : nullptr,
                                             g_w64.scalars + 1, g_w64.partials, nullptr);
    if (e != hipSuccess) return capi_hip_fail(e, "launch_f64_weighted_sumsq");
    HIP_TRY(hipDeviceSynchronize());   // dXe is released at the end of this block
  }
  HIP_TRY(hipDeviceSynchronize());
  int64_t nfail = 0;
  rsparse_hip_take_numeric_failures(&nfail, nullptr);
  double hs[2] = {0, 0};
  HIP_TRY(hipMemcpy(hs, g_w64.scalars, 2 * sizeof(double), hipMemcpyDeviceToHost));
  const double reg = (lambda > 0 && nx > 0) ? lambda * hs[1] : 0.0;
  if (ny) HIP_TRY(hipMemcpy(Y, dY.p, ny * 8, hipMemcpyDeviceToHost));
  if (loss_out) *loss_out = (hs[0] + reg) / (double)nnz;   // wrmf_implicit.hpp:304
  if (nfail)
    return fail(RSPARSE_HIP_ERR_NUMERIC, std::to_string(nfail) + " per-row systems were singular (not positive definite, and "
                                         "the general solver found a zero pivot column)");
  return RSPARSE_HIP_OK;
}

}  // namespace

extern "C" {

int rsparse_hip_csc_f64_create_device(int n_rows, int n_cols, const int32_t* d_col_ptrs, const int32_t* d_row_indices,
                                      const double* d_values, rsparse_hip_csc_f64** out) {
  if (!out) return fail(RSPARSE_HIP_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (n_rows < 0 || n_cols < 0) return fail(RSPARSE_HIP_ERR_INVALID, "negative matrix dimension");
  if (!d_col_ptrs) return fail(RSPARSE_HIP_ERR_INVALID, "col_ptrs is NULL");
  std::vector<int32_t> hp((size_t)n_cols + 1);
  HIP_TRY(hipMemcpy(hp.data(), d_col_ptrs, hp.size() * 4, hipMemcpyDeviceToHost));
  if (hp[0] != 0) return fail(RSPARSE_HIP_ERR_INVALID, "col_ptrs must start at 0");
  for (int c = 0; c < n_cols; c++)
    if (hp[(size_t)c + 1] < hp[(size_t)c]) return fail(RSPARSE_HIP_ERR_INVALID, "col_ptrs must be non-decreasing");
  const int64_t nnz = hp[(size_t)n_cols];
  if (nnz > 0 && (!d_row_indices || !d_values)) return fail(RSPARSE_HIP_ERR_INVALID, "row_indices or values is NULL");
  int bad = 0;   // the kernels gather X[row_index] unchecked
  HIP_TRY(check_row_indices_device(d_row_indices, nnz, n_rows, nullptr, &bad));
  if (bad) return fail(RSPARSE_HIP_ERR_INVALID, "row index out of range");
  rsparse_hip_csc_f64* m = new rsparse_hip_csc_f64();
  if (hipGetDevice(&m->device) != hipSuccess) {
    delete m;
    return fail(RSPARSE_HIP_ERR_RUNTIME, "no HIP device");
  }
  m->n_rows = n_rows; m->n_cols = n_cols; m->nnz = nnz;
  m->col_ptrs = d_col_ptrs; m->row_idx = d_row_indices; m->vals = const_cast<double*>(d_values);
  if (int rc = list_long_rows(*m, hp.data())) {
    delete m;
    return rc;
  }
  *out = m;
  return RSPARSE_HIP_OK;
}

int rsparse_hip_set_f64_long_rows(int min_len, int chunk_len) {
  if (min_len < 0 || chunk_len < 0) return fail(RSPARSE_HIP_ERR_INVALID, "lengths must be positive (0 = the default)");
  g_long_row_min = min_len > 0 ? min_len : kF64LongRowMin;
  g_long_row_chunk = chunk_len > 0 ? chunk_len : kF64LongRowChunk;
  return RSPARSE_HIP_OK;
}

int rsparse_hip_csc_f64_destroy(rsparse_hip_csc_f64* m) {
  delete m;
  return RSPARSE_HIP_OK;
}

int rsparse_hip_gramian_f64_device(const double* d_X, int rank, int64_t n, double lambda, double* d_XtX_out,
                                   double* d_sumsq_out, void* stream) {
  if (!d_X || !d_XtX_out) return fail(RSPARSE_HIP_ERR_INVALID, "X or XtX_out is NULL");
  if (rank <= 0 || n < 0) return fail(RSPARSE_HIP_ERR_INVALID, "rank must be positive and n non-negative");
  if (rank > RSPARSE_HIP_MAX_RANK_F64) return fail(RSPARSE_HIP_ERR_UNSUPPORTED, "rank > 128 is not on the fp64 device path");
  int rc = g_w64.ensure();
  if (rc) return rc;
  if ((rc = g_w64.ensure_gram(f64_gramian_scratch_doubles(rank)))) return rc;
  const double ridge = (double)(float)lambda;   // float::fl(diag(lambda)): rounded to fp32 in the double build too, R/model_WRMF.R:476
  hipError_t e = launch_f64_gramian(d_X, rank, n, ridge, d_XtX_out, d_sumsq_out, g_w64.gram, (hipStream_t)stream);
  if (e != hipSuccess) return capi_hip_fail(e, "launch_f64_gramian");
  return RSPARSE_HIP_OK;
}

int rsparse_hip_gramian_double(const double* X, int rank, int64_t n, double lambda, double* XtX_out) {
  if (!X || !XtX_out) return fail(RSPARSE_HIP_ERR_INVALID, "X or XtX_out is NULL");
  if (rank <= 0 || n < 0) return fail(RSPARSE_HIP_ERR_INVALID, "rank must be positive and n non-negative");
  DevBuf dX, dG;
  HIP_TRY(dX.alloc((size_t)rank * n * 8));
  HIP_TRY(dG.alloc((size_t)rank * rank * 8));
  if (n) HIP_TRY(hipMemcpy(dX.p, X, (size_t)rank * n * 8, hipMemcpyHostToDevice));
  int rc = rsparse_hip_gramian_f64_device(dX.as<double>(), rank, n, lambda, dG.as<double>(), nullptr, nullptr);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(XtX_out, dG.p, (size_t)rank * rank * 8, hipMemcpyDeviceToHost));
  return RSPARSE_HIP_OK;
}

int rsparse_hip_als_f64_device(const rsparse_hip_csc_f64* conf, int implicit, const double* d_X, double* d_Y,
                               const double* d_XtX, int rank, double lambda, unsigned solver, unsigned cg_steps,
                               int dynamic_lambda, int with_biases, int is_x_bias_last_row, double global_bias,
                               double* d_loss_rows_out, void* stream) {
  return f64_half_iteration(conf, implicit != 0, d_X, d_Y, d_XtX, rank, lambda, solver, cg_steps, dynamic_lambda,
                            with_biases, is_x_bias_last_row, global_bias, nullptr, nullptr, d_loss_rows_out,
                            (hipStream_t)stream);
}

int rsparse_hip_weighted_sumsq_f64_device(const double* d_X, int rank, int64_t n, const double* d_w, double* d_out,
                                          void* stream) {
  if (!d_X || !d_out) return fail(RSPARSE_HIP_ERR_INVALID, "X or out is NULL");
  if (rank <= 0 || n < 0) return fail(RSPARSE_HIP_ERR_INVALID, "rank must be positive and n non-negative");
  int rc = g_w64.ensure();
  if (rc) return rc;
  hipError_t e = launch_f64_weighted_sumsq(d_X, rank, n, d_w, d_out, g_w64.partials, (hipStream_t)stream);
  if (e != hipSuccess) return capi_hip_fail(e, "launch_f64_weighted_sumsq");
  return RSPARSE_HIP_OK;
}

int rsparse_hip_values_subtract_mean_f64_device(int64_t n, double* d_x, double* d_x_other, double* mean_out, void* stream) {
  if (n < 0) return fail(RSPARSE_HIP_ERR_INVALID, "negative length");
  if (mean_out) *mean_out = 0.0;
  if (n == 0) return RSPARSE_HIP_OK;
  if (!d_x) return fail(RSPARSE_HIP_ERR_INVALID, "values is NULL");
  hipStream_t s = (hipStream_t)stream;
  int rc = g_w64.ensure();
  if (rc) return rc;
  hipError_t e = launch_values_sum(d_x, n, g_w64.partials, g_w64.scalars + 2, s);
  if (e != hipSuccess) return capi_hip_fail(e, "launch_values_sum");
  const double inv = 1.0 / (double)n;
  if ((e = launch_values_subtract_mean(d_x, n, g_w64.scalars + 2, inv, s)) != hipSuccess)
    return capi_hip_fail(e, "launch_values_subtract_mean");
  if (d_x_other && (e = launch_values_subtract_mean(d_x_other, n, g_w64.scalars + 2, inv, s)) != hipSuccess)
    return capi_hip_fail(e, "launch_values_subtract_mean");
  double sum = 0.0;
  HIP_TRY(hipMemcpyAsync(&sum, g_w64.scalars + 2, sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (mean_out) *mean_out = sum * inv;
  return RSPARSE_HIP_OK;
}

int rsparse_hip_bias_sweep_explicit_f64_device(const rsparse_hip_csc_f64* conf, const double* d_other_bias, double lambda,
                                               int dynamic_lambda, int non_negative, double* d_out, void* stream) {
  if (!conf || !d_other_bias || !d_out) return fail(RSPARSE_HIP_ERR_INVALID, "NULL matrix or bias vector");
  hipError_t e = launch_bias_sweep(conf->col_ptrs, conf->row_idx, conf->vals, d_other_bias, conf->n_cols, lambda, dynamic_lambda,
                                   non_negative, d_out, (hipStream_t)stream);
  if (e != hipSuccess) return capi_hip_fail(e, "launch_bias_sweep");
  return RSPARSE_HIP_OK;
}

int rsparse_hip_bias_prep_implicit_f64_device(const rsparse_hip_csc_f64* conf, int n_other, double lambda, double* d_means,
                                              double* d_adj, void* stream) {
  if (!conf || !d_means || !d_adj) return fail(RSPARSE_HIP_ERR_INVALID, "NULL matrix or output");
  hipError_t e = launch_bias_implicit_prep(conf->col_ptrs, conf->vals, conf->n_cols, n_other, lambda, d_means, d_adj,
                                           (hipStream_t)stream);
  if (e != hipSuccess) return capi_hip_fail(e, "launch_bias_implicit_prep");
  return RSPARSE_HIP_OK;
}

int rsparse_hip_bias_sweep_implicit_f64_device(const rsparse_hip_csc_f64* conf, const double* d_other_bias, int n_other,
                                               const double* d_other_sum, const double* d_means, const double* d_adj,
                                               int non_negative, double global_bias, double* d_out, void* stream) {
  if (!conf || !d_other_bias || !d_means || !d_adj || !d_out) return fail(RSPARSE_HIP_ERR_INVALID, "NULL matrix or vector");
  hipError_t e = launch_bias_implicit_sweep(conf->col_ptrs, conf->row_idx, conf->vals, d_other_bias, conf->n_cols, n_other,
                                            d_other_sum, d_means, d_adj, non_negative, global_bias, d_out, (hipStream_t)stream);
  if (e != hipSuccess) return capi_hip_fail(e, "launch_bias_implicit_sweep");
  return RSPARSE_HIP_OK;
}

int rsparse_hip_initialize_biases_f64_device(rsparse_hip_csc_f64* c_ui, rsparse_hip_csc_f64* c_iu, double* d_user_bias,
                                             double* d_item_bias, double lambda, int dynamic_lambda, int non_negative,
                                             int calculate_global_bias, int is_explicit_feedback,
                                             double* global_bias_out, void* stream) {
  if (!c_ui || !c_iu || !d_user_bias || !d_item_bias) return fail(RSPARSE_HIP_ERR_INVALID, "NULL matrix or bias vector");
  const rsparse_hip_csc_f64& a = *c_ui;   // users x items, columns = items
  const rsparse_hip_csc_f64& b = *c_iu;   // items x users, columns = users
  if (a.n_rows != b.n_cols || a.n_cols != b.n_rows || a.nnz != b.nnz)
    return fail(RSPARSE_HIP_ERR_INVALID, "the two matrices are not transposes of each other");
  hipStream_t s = (hipStream_t)stream;
  int rc = g_w64.ensure();
  if (rc) return rc;
  const int n_items = a.n_cols, n_users = b.n_cols;
  double global_bias = 0.0;
  hipError_t e;
  if (is_explicit_feedback) {   // wrmf_utils.hpp:32-84
    if (calculate_global_bias && a.nnz > 0) {   // :41-52: mean of the values, removed from both orientations in place
      if ((e = launch_values_sum(a.vals, a.nnz, g_w64.partials, g_w64.scalars + 2, s)) != hipSuccess)
        return capi_hip_fail(e, "launch_values_sum");
      const double inv = 1.0 / (double)a.nnz;
      if ((e = launch_values_subtract_mean(a.vals, a.nnz, g_w64.scalars + 2, inv, s)) != hipSuccess ||
          (e = launch_values_subtract_mean(b.vals, b.nnz, g_w64.scalars + 2, inv, s)) != hipSuccess)
        return capi_hip_fail(e, "launch_values_subtract_mean");
      double sum = 0.0;
      HIP_TRY(hipMemcpyAsync(&sum, g_w64.scalars + 2, sizeof(double), hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      global_bias = sum * inv;
    }
    for (int iter = 0; iter < 5; iter++) {       // :54-82
      if ((e = launch_bias_sweep(a.col_ptrs, a.row_idx, a.vals, d_user_bias, a.n_cols, lambda, dynamic_lambda, non_negative,
                                 d_item_bias, s)) != hipSuccess ||
          (e = launch_bias_sweep(b.col_ptrs, b.row_idx, b.vals, d_item_bias, b.n_cols, lambda, dynamic_lambda, non_negative,
                                 d_user_bias, s)) != hipSuccess)
        return capi_hip_fail(e, "launch_bias_sweep");
    }
    if (global_bias_out) *global_bias_out = global_bias;
    return RSPARSE_HIP_OK;
  }
  // wrmf_utils.hpp:86-165
  DevBuf stats;   // means / adjustments of both sides (:97-124)
  HIP_TRY(stats.alloc(((size_t)2 * n_items + (size_t)2 * n_users + 4) * sizeof(double)));
  double* item_means = stats.as<double>();
  double* item_adj = item_means + n_items;
  double* user_means = item_adj + n_items;
  double* user_adj = user_means + n_users;
  if ((e = launch_bias_implicit_prep(a.col_ptrs, a.vals, n_items, n_users, lambda, item_means, item_adj, s)) != hipSuccess ||
      (e = launch_bias_implicit_prep(b.col_ptrs, b.vals, n_users, n_items, lambda, user_means, user_adj, s)) != hipSuccess)
    return capi_hip_fail(e, "launch_bias_implicit_prep");
  if (calculate_global_bias) {   // :90-93: sum(x) / (sum(x) + n_users n_items - nnz)
    if ((e = launch_values_sum(a.vals, a.nnz, g_w64.partials, g_w64.scalars + 2, s)) != hipSuccess)
      return capi_hip_fail(e, "launch_values_sum");
    double sum = 0.0;
    HIP_TRY(hipMemcpyAsync(&sum, g_w64.scalars + 2, sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    global_bias = sum / (sum + (double)n_users * (double)n_items - (double)a.nnz);
  }
  if (non_negative) global_bias = std::fmax(0.0, global_bias);
  if (global_bias_out) *global_bias_out = global_bias;
  for (int iter = 0; iter < 5; iter++) {   // :130-162
    const double* usum = nullptr;
    if (iter > 0) {                        // mean of the user biases of the previous sweep (:131-135)
      if ((e = launch_values_sum(d_user_bias, n_users, g_w64.partials, g_w64.scalars + 2, s)) != hipSuccess)
        return capi_hip_fail(e, "launch_values_sum");
      usum = g_w64.scalars + 2;
    }
    if ((e = launch_bias_implicit_sweep(a.col_ptrs, a.row_idx, a.vals, d_user_bias, n_items, n_users, usum, item_means,
                                        item_adj, non_negative, global_bias, d_item_bias, s)) != hipSuccess)
      return capi_hip_fail(e, "launch_bias_implicit_sweep");
    if ((e = launch_values_sum(d_item_bias, n_items, g_w64.partials, g_w64.scalars + 3, s)) != hipSuccess)
      return capi_hip_fail(e, "launch_values_sum");
    if ((e = launch_bias_implicit_sweep(b.col_ptrs, b.row_idx, b.vals, d_item_bias, n_users, n_items, g_w64.scalars + 3,
                                        user_means, user_adj, non_negative, global_bias, d_user_bias, s)) != hipSuccess)
      return capi_hip_fail(e, "launch_bias_implicit_sweep");
  }
  HIP_TRY(hipStreamSynchronize(s));   // `stats` is released on return
  return RSPARSE_HIP_OK;
}

int rsparse_hip_als_implicit_double(int n_rows, int n_cols, const int32_t* col_ptrs, const int32_t* row_indices,
                                    const double* values, const double* X, double* Y, const double* XtX, int rank,
                                    double lambda, int n_threads, unsigned solver, unsigned cg_steps,
                                    int with_biases, int is_x_bias_last_row, double global_bias,
                                    double* global_bias_base, int global_bias_base_len, int initialize_bias_base,
                                    double* loss_out) {
  (void)n_threads;
  return stateless_double(true, n_rows, n_cols, col_ptrs, row_indices, values, X, Y, XtX, nullptr, rank, lambda, solver,
                          cg_steps, 0, loss_out, with_biases, is_x_bias_last_row, global_bias, global_bias_base,
                          global_bias_base_len, initialize_bias_base);
}

int rsparse_hip_als_explicit_double(int n_rows, int n_cols, const int32_t* col_ptrs, const int32_t* row_indices,
                                    const double* values, const double* X, double* Y, const double* cnt_X, int rank,
                                    double lambda, unsigned n_threads, unsigned solver, unsigned cg_steps,
                                    int dynamic_lambda, int with_biases, int is_x_bias_last_row, double* loss_out) {
  (void)n_threads;
  return stateless_double(false, n_rows, n_cols, col_ptrs, row_indices, values, X, Y, nullptr, cnt_X, rank, lambda, solver,
                          cg_steps, dynamic_lambda, loss_out, with_biases, is_x_bias_last_row, 0.0, nullptr, 0, 1);
}

// replaces initialize_biases_double (src/wrmf_init.cpp:5-19): the bias vectors, the values and every sum in double
int rsparse_hip_initialize_biases_double(int n_users, int n_items, const int32_t* csc_p, const int32_t* csc_i,
                                         double* csc_x, const int32_t* csr_p, const int32_t* csr_i, double* csr_x,
                                         double* user_bias, double* item_bias, double lambda, int dynamic_lambda,
                                         int non_negative, int calculate_global_bias, int is_explicit_feedback,
                                         double* global_bias_out) {
  if (!user_bias || !item_bias) return fail(RSPARSE_HIP_ERR_INVALID, "user_bias or item_bias is NULL");
  if (n_users < 0 || n_items < 0 || !csc_p || !csr_p) return fail(RSPARSE_HIP_ERR_INVALID, "bad matrix");
  const int64_t nnz = csc_p[n_items];
  if (csr_p[n_users] != nnz) return fail(RSPARSE_HIP_ERR_INVALID, "the two matrices are not transposes of each other");
  if (nnz > 0 && (!csc_i || !csc_x || !csr_i || !csr_x)) return fail(RSPARSE_HIP_ERR_INVALID, "NULL index or value array");
  DevBuf p1, i1, x1, p2, i2, x2, dU, dI;
  HIP_TRY(p1.alloc(((size_t)n_items + 1) * 4)); HIP_TRY(i1.alloc((size_t)nnz * 4)); HIP_TRY(x1.alloc((size_t)nnz * 8));
  HIP_TRY(p2.alloc(((size_t)n_users + 1) * 4)); HIP_TRY(i2.alloc((size_t)nnz * 4)); HIP_TRY(x2.alloc((size_t)nnz * 8));
  HIP_TRY(dU.alloc((size_t)n_users * 8)); HIP_TRY(dI.alloc((size_t)n_items * 8));
  HIP_TRY(hipMemcpy(p1.p, csc_p, ((size_t)n_items + 1) * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(p2.p, csr_p, ((size_t)n_users + 1) * 4, hipMemcpyHostToDevice));
  if (nnz) {
    HIP_TRY(hipMemcpy(i1.p, csc_i, (size_t)nnz * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(x1.p, csc_x, (size_t)nnz * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(i2.p, csr_i, (size_t)nnz * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(x2.p, csr_x, (size_t)nnz * 8, hipMemcpyHostToDevice));
  }
  if (n_users) HIP_TRY(hipMemcpy(dU.p, user_bias, (size_t)n_users * 8, hipMemcpyHostToDevice));
  if (n_items) HIP_TRY(hipMemcpy(dI.p, item_bias, (size_t)n_items * 8, hipMemcpyHostToDevice));
  rsparse_hip_csc_f64 *c_ui = nullptr, *c_iu = nullptr;
  int rc = rsparse_hip_csc_f64_create_device(n_users, n_items, p1.as<int32_t>(), i1.as<int32_t>(), x1.as<double>(), &c_ui);
  if (rc) return rc;
  struct Guard { rsparse_hip_csc_f64* c; ~Guard() { rsparse_hip_csc_f64_destroy(c); } } g1{c_ui};
  if ((rc = rsparse_hip_csc_f64_create_device(n_items, n_users, p2.as<int32_t>(), i2.as<int32_t>(), x2.as<double>(), &c_iu)))
    return rc;
  Guard g2{c_iu};
  double gb = 0.0;
  rc = rsparse_hip_initialize_biases_f64_device(c_ui, c_iu, dU.as<double>(), dI.as<double>(), lambda, dynamic_lambda,
                                                non_negative, calculate_global_bias, is_explicit_feedback, &gb, nullptr);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  if (n_users) HIP_TRY(hipMemcpy(user_bias, dU.p, (size_t)n_users * 8, hipMemcpyDeviceToHost));
  if (n_items) HIP_TRY(hipMemcpy(item_bias, dI.p, (size_t)n_items * 8, hipMemcpyDeviceToHost));
  if (is_explicit_feedback && calculate_global_bias && nnz) {
    // the reference removes the global mean from the @x slots of BOTH matrices in place (wrmf_utils.hpp:41-52)
    HIP_TRY(hipMemcpy(csc_x, x1.p, (size_t)nnz * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(csr_x, x2.p, (size_t)nnz * 8, hipMemcpyDeviceToHost));
  }
  if (global_bias_out) *global_bias_out = gb;
  return RSPARSE_HIP_OK;
}

}  // extern "C"
