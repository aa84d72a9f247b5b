// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement of rsparse's WRMF/ALS hot path, used as (i) the parity checker for the HIP
// kernels and (ii) the timed "rsparse-shaped OpenMP" CPU baseline in bench.py (cpu_baseline.kind
// = "port").  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
//
// PARITY UNPINNED: the reference itself cannot be built or run here (needs R, Rcpp,
// RcppArmadillo + BLAS/LAPACK, none present; its tests hold no numeric golden vectors for this
// path -- tests/testthat/test-wrmf.R asserts shapes only).  This restatement is therefore pinned
// by closed-form identities instead (tests/test_oracle.py): dense numpy.linalg.solve of the stated
// normal equations, CG(cg_steps>=k) -> Cholesky answer, fit_transform == transform.
//
// Every function cites the reference lines it follows (paths relative to /root/reference).
// Plain C++17 + OpenMP, no third-party dependency: Armadillo expressions are written out as loops.
//
// Layouts (src/utils.cpp:69-78, inst/include/mapped_csc.hpp:8-29):
//   Conf : CSC, col_ptrs int32[n_cols+1], row_indices int32[nnz] (0-based), values f64[nnz]
//   X    : rank x n_rows(Conf)   column-major (each entity's rank-vector contiguous), read only
//   Y    : rank x n_cols(Conf)   column-major, in: CG warm start, out: solution
//   XtX  : rank x rank, already contains + lambda*I (R/model_WRMF.R:474-486)

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// inst/include/wrmf.hpp:16-22
constexpr unsigned CHOLESKY = 0;
constexpr unsigned CONJUGATE_GRADIENT = 1;
constexpr unsigned SEQ_COORDINATE_WISE_NNLS = 2;
constexpr unsigned SCD_MAX_ITER = 10000;
constexpr double SCD_TOL = 1e-4;
constexpr double CG_TOL = 1e-10;
constexpr double NNLS_EPS = 1e-16;   // EPS, inst/include/nnls.hpp:8

template <class T>
inline T dot(const T* a, const T* b, int n) {
  // arma::dot on short vectors: two running sums in T (op_dot::direct_dot_arma)
  T s1 = 0, s2 = 0;
  int i = 0;
  for (; i + 1 < n; i += 2) {
    s1 += a[i] * b[i];
    s2 += a[i + 1] * b[i + 1];
  }
  if (i < n) s1 += a[i] * b[i];
  return s1 + s2;
}

// y = A * x, A is k x k column-major (XtX * v, wrmf_implicit.hpp:16,22)
template <class T>
inline void gemv_sq(const T* A, const T* x, T* y, int k) {
  for (int r = 0; r < k; r++) y[r] = 0;
  for (int c = 0; c < k; c++) {
    const T xc = x[c];
    const T* col = A + (size_t)c * k;
    for (int r = 0; r < k; r++) y[r] += col[r] * xc;
  }
}

// t = Xnnz^T * v  (n-vector); Xnnz is k x n column-major
template <class T>
inline void gemv_t(const T* Xn, const T* v, T* t, int k, int n) {
  for (int j = 0; j < n; j++) t[j] = dot(Xn + (size_t)j * k, v, k);
}

// u (+)= Xnnz * w  (k-vector)
template <class T>
inline void gemv_n(const T* Xn, const T* w, T* u, int k, int n, bool accumulate) {
  if (!accumulate)
    for (int r = 0; r < k; r++) u[r] = 0;
  for (int j = 0; j < n; j++) {
    const T wj = w[j];
    const T* col = Xn + (size_t)j * k;
    for (int r = 0; r < k; r++) u[r] += col[r] * wj;
  }
}

template <class T>
struct Scratch {
  std::vector<T> Xn, conf, t, w, x, r, p, Ap, tmp, lhs, rhs;
  void ensure(int k, int n) {
    if ((size_t)k * n > Xn.size()) Xn.resize((size_t)k * n);
    if ((size_t)n > conf.size()) { conf.resize(n); t.resize(n); w.resize(n); }
    if ((size_t)k > x.size()) {
      x.resize(k); r.resize(k); p.resize(k); Ap.resize(k); tmp.resize(k); rhs.resize(k);
      lhs.resize((size_t)k * k);
    }
  }
};

// inst/include/wrmf_implicit.hpp:8-32  cg_solver_implicit<T>
//   r = X_nnz*(c - c1 % (X_nnz^T x)) - XtX*x ; p = r ; rsold = r.r           (:16-19)
//   loop: Ap = XtX*p + X_nnz*(c1 % (X_nnz^T p)); alpha = rsold/(p.Ap);       (:22-23)
//         x += alpha p; r -= alpha Ap; rsnew = r.r; if rsnew < CG_TOL break; (:24-27)
//         p = r + p*(rsnew/rsold); rsold = rsnew                             (:28-29)
// rsold/rsnew/alpha are double even for T=float (:18); a double scalar times an
// arma::Col<float> is rounded to float first (elem_type conversion).
template <class T>
void cg_solver_implicit(Scratch<T>& s, int k, int n, const T* XtX, unsigned n_iter) {
  T* x = s.x.data(); T* r = s.r.data(); T* p = s.p.data(); T* Ap = s.Ap.data();
  T* t = s.t.data(); T* w = s.w.data(); T* tmp = s.tmp.data();
  const T* Xn = s.Xn.data(); const T* c = s.conf.data();
  gemv_t(Xn, x, t, k, n);
  for (int j = 0; j < n; j++) w[j] = c[j] - (c[j] - (T)1.0) * t[j];
  gemv_n(Xn, w, r, k, n, false);
  gemv_sq(XtX, x, tmp, k);
  for (int i = 0; i < k; i++) { r[i] -= tmp[i]; p[i] = r[i]; }
  double rsold = dot(r, r, k), rsnew, alpha;
  for (unsigned it = 0; it < n_iter; it++) {
    gemv_sq(XtX, p, Ap, k);
    gemv_t(Xn, p, t, k, n);
    for (int j = 0; j < n; j++) w[j] = (c[j] - (T)1.0) * t[j];
    gemv_n(Xn, w, Ap, k, n, true);
    alpha = rsold / dot(p, Ap, k);
    const T a = (T)alpha;
    for (int i = 0; i < k; i++) { x[i] += a * p[i]; r[i] -= a * Ap[i]; }
    rsnew = dot(r, r, k);
    if (rsnew < CG_TOL) break;
    const T b = (T)(rsnew / rsold);
    for (int i = 0; i < k; i++) p[i] = r[i] + p[i] * b;
    rsold = rsnew;
  }
}

// inst/include/wrmf_implicit.hpp:35-57  cg_solver_implicit_global_bias<T> (call site :203)
//   r = X_nnz*(c - c1 % (X_nnz^T x + global_bias)) - XtX*x + global_bias_base ; the loop is cg_solver_implicit's (:46-55)
// (marked "very poor numerical precision" at :34 -- restated as written).  `n` may be 0: with a global bias every column
// is solved (:178), an empty one against  r = global_bias_base - XtX*x.
template <class T>
void cg_solver_implicit_global_bias(Scratch<T>& s, int k, int n, const T* XtX, unsigned n_iter, const T* base, T gbias) {
  T* x = s.x.data(); T* r = s.r.data(); T* p = s.p.data(); T* Ap = s.Ap.data();
  T* t = s.t.data(); T* w = s.w.data(); T* tmp = s.tmp.data();
  const T* Xn = s.Xn.data(); const T* c = s.conf.data();
  gemv_t(Xn, x, t, k, n);
  for (int j = 0; j < n; j++) w[j] = c[j] - (c[j] - (T)1.0) * (t[j] + gbias);
  gemv_n(Xn, w, r, k, n, false);
  gemv_sq(XtX, x, tmp, k);
  for (int i = 0; i < k; i++) { r[i] = (r[i] - tmp[i]) + base[i]; p[i] = r[i]; }
  double rsold = dot(r, r, k), rsnew, alpha;
  for (unsigned it = 0; it < n_iter; it++) {
    gemv_sq(XtX, p, Ap, k);
    gemv_t(Xn, p, t, k, n);
    for (int j = 0; j < n; j++) w[j] = (c[j] - (T)1.0) * t[j];
    gemv_n(Xn, w, Ap, k, n, true);
    alpha = rsold / dot(p, Ap, k);
    const T a = (T)alpha;
    for (int i = 0; i < k; i++) { x[i] += a * p[i]; r[i] -= a * Ap[i]; }
    rsnew = dot(r, r, k);
    if (rsnew < CG_TOL) break;
    const T b = (T)(rsnew / rsold);
    for (int i = 0; i < k; i++) p[i] = r[i] + p[i] * b;
    rsold = rsnew;
  }
}

// inst/include/wrmf_explicit.hpp:8-31  cg_solver_explicit<T>
//   r = X_nnz*(c - X_nnz^T x) - lambda x ; Ap = X_nnz*(X_nnz^T p) + lambda p  (:15,21)
template <class T>
void cg_solver_explicit(Scratch<T>& s, int k, int n, T lambda, unsigned n_iter) {
  T* x = s.x.data(); T* r = s.r.data(); T* p = s.p.data(); T* Ap = s.Ap.data();
  T* t = s.t.data(); T* w = s.w.data();
  const T* Xn = s.Xn.data(); const T* c = s.conf.data();
  gemv_t(Xn, x, t, k, n);
  for (int j = 0; j < n; j++) w[j] = c[j] - t[j];
  gemv_n(Xn, w, r, k, n, false);
  for (int i = 0; i < k; i++) { r[i] -= lambda * x[i]; p[i] = r[i]; }
  double rsold = dot(r, r, k), rsnew, alpha;
  for (unsigned it = 0; it < n_iter; it++) {
    gemv_t(Xn, p, t, k, n);
    gemv_n(Xn, t, Ap, k, n, false);
    for (int i = 0; i < k; i++) Ap[i] += lambda * p[i];
    alpha = rsold / dot(p, Ap, k);
    const T a = (T)alpha;
    for (int i = 0; i < k; i++) { x[i] += a * p[i]; r[i] -= a * Ap[i]; }
    rsnew = dot(r, r, k);
    if (rsnew < CG_TOL) break;
    const T b = (T)(rsnew / rsold);
    for (int i = 0; i < k; i++) p[i] = r[i] + p[i] * b;
    rsold = rsnew;
  }
}

// arma::solve(lhs, rhs, fast [+ likely_sympd]) (wrmf_implicit.hpp:236, wrmf_explicit.hpp:108):
// LAPACK posv (Cholesky LL^T, no refinement); if the factorisation fails Armadillo falls back to
// a general solver -- restated here as LU with partial pivoting (gesv).  A is k x k column-major,
// overwritten; b overwritten with the solution.  Returns false if singular.
template <class T>
bool solve_sympd(T* A, T* b, int k, std::vector<T>& keep) {
  keep.assign(A, A + (size_t)k * k);
  bool ok = true;
  for (int j = 0; j < k && ok; j++) {
    T d = A[(size_t)j * k + j];
    for (int m = 0; m < j; m++) d -= A[(size_t)m * k + j] * A[(size_t)m * k + j];
    if (!(d > 0)) { ok = false; break; }
    d = std::sqrt(d);
    A[(size_t)j * k + j] = d;
    for (int i = j + 1; i < k; i++) {
      T v = A[(size_t)j * k + i];
      for (int m = 0; m < j; m++) v -= A[(size_t)m * k + i] * A[(size_t)m * k + j];
      A[(size_t)j * k + i] = v / d;
    }
  }
  if (ok) {
    for (int i = 0; i < k; i++) {  // L z = b
      T v = b[i];
      for (int m = 0; m < i; m++) v -= A[(size_t)m * k + i] * b[m];
      b[i] = v / A[(size_t)i * k + i];
    }
    for (int i = k - 1; i >= 0; i--) {  // L^T y = z
      T v = b[i];
      for (int m = i + 1; m < k; m++) v -= A[(size_t)i * k + m] * b[m];
      b[i] = v / A[(size_t)i * k + i];
    }
    return true;
  }
  // gesv fallback
  std::copy(keep.begin(), keep.end(), A);
  for (int c = 0; c < k; c++) {
    int piv = c;
    T best = std::fabs(A[(size_t)c * k + c]);
    for (int i = c + 1; i < k; i++)
      if (std::fabs(A[(size_t)c * k + i]) > best) { best = std::fabs(A[(size_t)c * k + i]); piv = i; }
    if (best == 0) return false;
    if (piv != c) {
      for (int m = 0; m < k; m++) std::swap(A[(size_t)m * k + c], A[(size_t)m * k + piv]);
      std::swap(b[c], b[piv]);
    }
    for (int i = c + 1; i < k; i++) {
      const T f = A[(size_t)c * k + i] / A[(size_t)c * k + c];
      if (f == 0) continue;
      for (int m = c; m < k; m++) A[(size_t)m * k + i] -= f * A[(size_t)m * k + c];
      b[i] -= f * b[c];
    }
  }
  for (int i = k - 1; i >= 0; i--) {
    T v = b[i];
    for (int m = i + 1; m < k; m++) v -= A[(size_t)m * k + i] * b[m];
    b[i] = v / A[(size_t)i * k + i];
  }
  return true;
}

// arma::accu(X % X) for a dense matrix: two running sums in T over consecutive pairs.
template <class T>
T accu_sq(const T* X, size_t n) {
  T a = 0, b = 0;
  size_t i = 0;
  for (; i + 1 < n; i += 2) { a += X[i] * X[i]; b += X[i + 1] * X[i + 1]; }
  if (i < n) a += X[i] * X[i];
  return a + b;
}

// inst/include/wrmf_implicit.hpp:90-305, no-bias / no-global-bias branch only
// (:160-185 column loop, :195-197 CG, :206-208,231,236 Cholesky, :254 write-back,
//  :259-261 loss, :272-283 empty column -> zeros, :286-304 + lambda*accu(X%X), / nnz).
// inst/include/nnls.hpp:10-48 -- c_nnls(X = lhs, y = rhs, init): XtX = X^T X (+EPS on the diagonal),
// mu = XtX init - X^T y, then scd_ls_update: sweeps over the coordinates in order,
// new = max(0, h_k - mu_k / XtX_kk), mu += (new - h_k) XtX[:,k], until the largest relative coordinate
// change of a sweep is <= rel_tol or max_iter sweeps.  X is k x k column-major; h holds init on entry.
template <class T>
void c_nnls(const T* X, const T* y, T* h, int k, unsigned max_iter, double rel_tol, std::vector<T>& work) {
  work.resize((size_t)k * k + (size_t)k);
  T* XtX = work.data();
  T* mu = XtX + (size_t)k * k;
  for (int c = 0; c < k; c++)
    for (int r = 0; r < k; r++) {
      T s = 0;
      for (int m = 0; m < k; m++) s += X[(size_t)r * k + m] * X[(size_t)c * k + m];   // (X^T X)[r][c]
      XtX[(size_t)c * k + r] = s;
    }
  for (int d = 0; d < k; d++) XtX[(size_t)d * k + d] += (T)NNLS_EPS;                     // :44
  for (int r = 0; r < k; r++) {                                                          // :45
    T s = 0;
    for (int c = 0; c < k; c++) s += XtX[(size_t)c * k + r] * h[c];
    T xty = 0;
    for (int m = 0; m < k; m++) xty += X[(size_t)r * k + m] * y[m];
    mu[r] = s - xty;
  }
  for (unsigned t = 0; t < max_iter; t++) {                                              // :17-34
    T rel_diff = 0;
    for (int c = 0; c < k; c++) {
      const T old_value = h[c];
      T new_value = old_value - mu[c] / XtX[(size_t)c * k + c];
      if (new_value < 0) new_value = 0;
      const T diff = new_value - old_value;
      if (diff != 0) {
        h[c] = new_value;
        const T* col = XtX + (size_t)c * k;
        for (int r = 0; r < k; r++) mu[r] += diff * col[r];
        const double step_err = std::fabs((double)diff) / (std::fabs((double)old_value) + NNLS_EPS);
        if (step_err > rel_diff) rel_diff = (T)step_err;
      }
    }
    if (rel_diff <= rel_tol) break;
  }
}

template <class T>
double als_implicit(int n_rows, int n_cols, const int32_t* col_ptrs, const int32_t* row_indices,
                    const double* values, const T* X, T* Y, const T* XtX, int k, double lambda,
                    int n_threads, unsigned solver, unsigned cg_steps, int* status) {
  double loss = 0;
  int bad = 0;
  const size_t nnz = (size_t)col_ptrs[n_cols];
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
  {
    Scratch<T> s;
    std::vector<T> keep;
#pragma omp for schedule(dynamic) reduction(+ : loss) reduction(+ : bad)
    for (int i = 0; i < n_cols; i++) {
      const int p1 = col_ptrs[i], p2 = col_ptrs[i + 1];
      T* y = Y + (size_t)i * k;
      if (p1 < p2) {
        const int n = p2 - p1;
        s.ensure(k, n);
        for (int j = 0; j < n; j++) {
          s.conf[j] = (T)values[p1 + j];                                  // :182-183 conv_to<T>
          std::memcpy(&s.Xn[(size_t)j * k], X + (size_t)row_indices[p1 + j] * k, sizeof(T) * k);  // :184
        }
        if (solver == CONJUGATE_GRADIENT) {
          std::memcpy(s.x.data(), y, sizeof(T) * k);                      // :185 warm start
          cg_solver_implicit<T>(s, k, n, XtX, cg_steps);
        } else {
          // lhs = XtX + (X_nnz.each_row() % (c-1)^T) * X_nnz^T ; rhs = X_nnz * c   (:207-208,231)
          T* lhs = s.lhs.data();
          std::memcpy(lhs, XtX, sizeof(T) * k * k);
          for (int j = 0; j < n; j++) {
            const T c1 = s.conf[j] - (T)1.0;
            const T* col = &s.Xn[(size_t)j * k];
            for (int b = 0; b < k; b++) {
              const T f = col[b] * c1;
              for (int a = 0; a < k; a++) lhs[(size_t)b * k + a] += col[a] * f;
            }
          }
          gemv_n(s.Xn.data(), s.conf.data(), s.x.data(), k, n, false);
          if (solver == SEQ_COORDINATE_WISE_NNLS) {                       // :233-234
            std::vector<T> rhs(s.x.begin(), s.x.begin() + k);
            std::memcpy(s.x.data(), y, sizeof(T) * k);                    // init = Y.col(i)
            c_nnls<T>(lhs, rhs.data(), s.x.data(), k, SCD_MAX_ITER, SCD_TOL, keep);
          } else if (!solve_sympd(lhs, s.x.data(), k, keep)) {
            bad += 1;
          }
        }
        std::memcpy(y, s.x.data(), sizeof(T) * k);                        // :254
        // loss += dot(square(1 - y^T X_nnz), c) + lambda * dot(y, y)      (:259-261)
        gemv_t(s.Xn.data(), s.x.data(), s.t.data(), k, n);
        T l = 0;
        for (int j = 0; j < n; j++) {
          const T d = (T)1.0 - s.t[j];
          l += d * d * s.conf[j];
        }
        loss += l + lambda * dot(s.x.data(), s.x.data(), k);
      } else {
        for (int r = 0; r < k; r++) y[r] = 0;                             // :281
      }
    }
  }
  if (lambda > 0) loss += lambda * accu_sq(X, (size_t)k * n_rows);        // :286-301
  if (status) *status = bad;
  return loss / (double)nnz;                                              // :304
}

// inst/include/wrmf_explicit.hpp:33-174, no-bias branch (:68-109 column loop, :78 lambda_use,
// :103-108 Cholesky, :131-132 loss, :133-144 empty column, :146-173 regulariser on X, / nnz).
template <class T>
double als_explicit(int n_rows, int n_cols, const int32_t* col_ptrs, const int32_t* row_indices,
                    const double* values, const T* X, T* Y, const T* cnt_X, int k, double lambda,
                    int n_threads, unsigned solver, unsigned cg_steps, int dynamic_lambda,
                    int* status) {
  double loss = 0;
  int bad = 0;
  const size_t nnz = (size_t)col_ptrs[n_cols];
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
  {
    Scratch<T> s;
    std::vector<T> keep;
#pragma omp for schedule(dynamic, 100) reduction(+ : loss) reduction(+ : bad)
    for (int i = 0; i < n_cols; i++) {
      const int p1 = col_ptrs[i], p2 = col_ptrs[i + 1];
      T* y = Y + (size_t)i * k;
      if (p1 < p2) {
        const int n = p2 - p1;
        s.ensure(k, n);
        const T lambda_use = (T)(lambda * (dynamic_lambda ? (double)(T)n : 1.0));   // :78
        for (int j = 0; j < n; j++) {
          s.conf[j] = (T)values[p1 + j];
          std::memcpy(&s.Xn[(size_t)j * k], X + (size_t)row_indices[p1 + j] * k, sizeof(T) * k);
        }
        if (solver == CONJUGATE_GRADIENT) {
          std::memcpy(s.x.data(), y, sizeof(T) * k);
          cg_solver_explicit<T>(s, k, n, lambda_use, cg_steps);
        } else {
          T* lhs = s.lhs.data();                                          // :103-105
          std::fill(lhs, lhs + (size_t)k * k, (T)0);
          for (int j = 0; j < n; j++) {
            const T* col = &s.Xn[(size_t)j * k];
            for (int b = 0; b < k; b++) {
              const T f = col[b];
              for (int a = 0; a < k; a++) lhs[(size_t)b * k + a] += col[a] * f;
            }
          }
          for (int a = 0; a < k; a++) lhs[(size_t)a * k + a] += lambda_use;
          gemv_n(s.Xn.data(), s.conf.data(), s.x.data(), k, n, false);
          if (solver == SEQ_COORDINATE_WISE_NNLS) {                       // :109-110
            std::vector<T> rhs(s.x.begin(), s.x.begin() + k);
            std::memcpy(s.x.data(), y, sizeof(T) * k);
            c_nnls<T>(lhs, rhs.data(), s.x.data(), k, SCD_MAX_ITER, SCD_TOL, keep);
          } else if (!solve_sympd(lhs, s.x.data(), k, keep)) {
            bad += 1;
          }
        }
        std::memcpy(y, s.x.data(), sizeof(T) * k);
        gemv_t(s.Xn.data(), s.x.data(), s.t.data(), k, n);                // :131-132
        T l = 0;
        for (int j = 0; j < n; j++) {
          const T d = s.conf[j] - s.t[j];
          l += d * d;
        }
        loss += l + lambda_use * dot(s.x.data(), s.x.data(), k);
      } else {
        for (int r = 0; r < k; r++) y[r] = 0;
      }
    }
  }
  if (lambda > 0) {                                                       // :160-170
    if (!dynamic_lambda) {
      loss += lambda * accu_sq(X, (size_t)k * n_rows);
    } else {
      // accu((X % X) * cnt_X): row sums weighted by cnt_X, then summed
      T tot = 0;
      std::vector<T> rowacc(k, (T)0);
      for (int j = 0; j < n_rows; j++) {
        const T* col = X + (size_t)j * k;
        const T cj = cnt_X ? cnt_X[j] : (T)0;
        for (int r = 0; r < k; r++) rowacc[r] += col[r] * col[r] * cj;
      }
      for (int r = 0; r < k; r++) tot += rowacc[r];
      loss += lambda * tot;
    }
  }
  if (status) *status = bad;
  return loss / (double)nnz;
}

// inst/include/wrmf_explicit.hpp:33-174, with_biases branch: the model has two extra coordinates, a column of
// ones that multiplies the other side's bias and the own bias.  X = [1, ..., x_bias] / Y = [y_bias, ..., 1] when
// is_x_bias_last_row, X = [x_bias, ..., 1] / Y = [1, ..., y_bias] otherwise (:41-57).  Per row: X_nnz loses its
// x_bias row (:88 drop_row(X_nnz, is_x_bias_last_row)), the ratings lose the fixed side's biases (:89), the
// warm start loses a row too -- drop_row(init, !is_x_bias_last_row) (:90), i.e. the FIRST entry when the x bias
// is last -- the (rank-1)-vector is solved as in the no-bias branch and written to head/tail of Y.col(i)
// (:115-127).  The regulariser on X skips the row of ones (:147-159).
template <class T>
double als_explicit_biases(int n_rows, int n_cols, const int32_t* col_ptrs, const int32_t* row_indices,
                           const double* values, const T* X, T* Y, const T* cnt_X, int k, double lambda,
                           int n_threads, unsigned solver, unsigned cg_steps, int dynamic_lambda,
                           int is_x_bias_last_row, int* status) {
  double loss = 0;
  int bad = 0;
  const size_t nnz = (size_t)col_ptrs[n_cols];
  const int k1 = k - 1;
  const int xoff = is_x_bias_last_row ? 0 : 1;      // first kept row of X_nnz
  const int xb = is_x_bias_last_row ? k - 1 : 0;    // row of X holding x_biases (:59-64)
  const int ioff = is_x_bias_last_row ? 1 : 0;      // first kept entry of the warm start (:90)
  const int ooff = is_x_bias_last_row ? 0 : 1;      // head / tail of Y.col(i) (:115-127)
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
  {
    Scratch<T> s;
    std::vector<T> keep;
#pragma omp for schedule(dynamic, 100) reduction(+ : loss) reduction(+ : bad)
    for (int i = 0; i < n_cols; i++) {
      const int p1 = col_ptrs[i], p2 = col_ptrs[i + 1];
      T* y = Y + (size_t)i * k;
      if (p1 < p2) {
        const int n = p2 - p1;
        s.ensure(k1, n);
        const T lambda_use = (T)(lambda * (dynamic_lambda ? (double)(T)n : 1.0));   // :78
        for (int j = 0; j < n; j++) {
          const T* xc = X + (size_t)row_indices[p1 + j] * k;
          s.conf[j] = (T)values[p1 + j] - xc[xb];                                    // :89
          std::memcpy(&s.Xn[(size_t)j * k1], xc + xoff, sizeof(T) * k1);             // :88
        }
        std::memcpy(s.x.data(), y + ioff, sizeof(T) * k1);                           // :90
        if (solver == CONJUGATE_GRADIENT) {
          cg_solver_explicit<T>(s, k1, n, lambda_use, cg_steps);
        } else {
          T* lhs = s.lhs.data();
          std::fill(lhs, lhs + (size_t)k1 * k1, (T)0);
          for (int j = 0; j < n; j++) {
            const T* col = &s.Xn[(size_t)j * k1];
            for (int b = 0; b < k1; b++) {
              const T f = col[b];
              for (int a = 0; a < k1; a++) lhs[(size_t)b * k1 + a] += col[a] * f;
            }
          }
          for (int a = 0; a < k1; a++) lhs[(size_t)a * k1 + a] += lambda_use;
          std::vector<T> init(s.x.begin(), s.x.begin() + k1);
          gemv_n(s.Xn.data(), s.conf.data(), s.x.data(), k1, n, false);
          if (solver == SEQ_COORDINATE_WISE_NNLS) {
            std::vector<T> rhs(s.x.begin(), s.x.begin() + k1);
            std::memcpy(s.x.data(), init.data(), sizeof(T) * k1);
            c_nnls<T>(lhs, rhs.data(), s.x.data(), k1, SCD_MAX_ITER, SCD_TOL, keep);
          } else if (!solve_sympd(lhs, s.x.data(), k1, keep)) {
            bad += 1;
          }
        }
        std::memcpy(y + ooff, s.x.data(), sizeof(T) * k1);
        gemv_t(s.Xn.data(), s.x.data(), s.t.data(), k1, n);                          // :131-132
        T l = 0;
        for (int j = 0; j < n; j++) {
          const T d = s.conf[j] - s.t[j];
          l += d * d;
        }
        loss += l + lambda_use * dot(s.x.data(), s.x.data(), k1);
      } else {
        for (int r = 0; r < k1; r++) y[ooff + r] = 0;                                // :134-141
      }
    }
  }
  if (lambda > 0) {                                                                  // :147-159
    const int r0 = is_x_bias_last_row ? 1 : 0, r1 = is_x_bias_last_row ? k : k - 1;  // all rows but the ones
    T tot = 0;
    std::vector<T> rowacc(k, (T)0);
    for (int j = 0; j < n_rows; j++) {
      const T* col = X + (size_t)j * k;
      const T cj = dynamic_lambda ? (cnt_X ? cnt_X[j] : (T)0) : (T)1;
      for (int r = r0; r < r1; r++) rowacc[r] += col[r] * col[r] * cj;
    }
    for (int r = r0; r < r1; r++) tot += rowacc[r];
    loss += lambda * tot;
  }
  if (status) *status = bad;
  return loss / (double)nnz;
}

// inst/include/wrmf_implicit.hpp:90-305, with_biases branch without global bias, Cholesky and NNLS solvers (the
// CG + bias combination drops a row of the warm start twice, :197, and cannot run in the reference).
// XtX is (rank-1) x (rank-1): tcrossprod of X without its x_bias row + ridge (R/model_WRMF.R:474-486).
//   x_biases = the bias row of X (:116-120);  rhs_init = -drop_row(X) * x_biases (:142-147)
//   every row is solved, empty ones too (:178);  lhs = XtX + X_nnz diag(c-1) X_nnz^T (:207-208)
//   rhs = rhs_init + X_nnz (c - x_biases(idx) % (c - 1)) (:226);  head / tail of Y.col(i) (:240-252)
//   loss: dot(square(1 - y^T X_nnz - x_biases(idx)), c) + lambda y.y, or lambda y.y for an empty row (:256-270)
//   regulariser lambda * accu(X without the row of ones ^2) (:287-297)
template <class T>
double als_implicit_biases(int n_rows, int n_cols, const int32_t* col_ptrs, const int32_t* row_indices,
                           const double* values, const T* X, T* Y, const T* XtX, int k, double lambda,
                           int n_threads, unsigned solver, int is_x_bias_last_row, int* status,
                           double global_bias_in = 0.0) {
  double loss = 0;
  int bad = 0;
  // :108-109 -- a global bias below sqrt(eps) counts as zero; with one, rhs_init = -X' (x_b + global_bias) (:152) and
  // the loss compares with 1 - global_bias - x_b (:268-270)
  const T gbias = global_bias_in < std::sqrt((double)std::numeric_limits<T>::epsilon()) ? (T)0 : (T)global_bias_in;
  const size_t nnz = (size_t)col_ptrs[n_cols];
  const int k1 = k - 1;
  const int xoff = is_x_bias_last_row ? 0 : 1;
  const int xb = is_x_bias_last_row ? k - 1 : 0;
  const int ioff = is_x_bias_last_row ? 1 : 0;      // drop_row(init, !is_x_bias_last_row) (:189)
  const int ooff = is_x_bias_last_row ? 0 : 1;
  std::vector<T> rhs_init(k1, (T)0);
  for (int j = 0; j < n_rows; j++) {
    const T* xc = X + (size_t)j * k;
    const T b = xc[xb] + gbias;
    for (int r = 0; r < k1; r++) rhs_init[r] -= xc[xoff + r] * b;
  }
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
  {
    Scratch<T> s;
    std::vector<T> keep, xbn;
#pragma omp for schedule(dynamic) reduction(+ : loss) reduction(+ : bad)
    for (int i = 0; i < n_cols; i++) {
      const int p1 = col_ptrs[i], p2 = col_ptrs[i + 1];
      T* y = Y + (size_t)i * k;
      const int n = p2 - p1;
      s.ensure(k1, n > 0 ? n : 1);
      xbn.resize(n > 0 ? n : 1);
      for (int j = 0; j < n; j++) {
        const T* xc = X + (size_t)row_indices[p1 + j] * k;
        s.conf[j] = (T)values[p1 + j];
        xbn[j] = xc[xb];
        std::memcpy(&s.Xn[(size_t)j * k1], xc + xoff, sizeof(T) * k1);
      }
      std::vector<T> init(y + ioff, y + ioff + k1);
      T* lhs = s.lhs.data();
      std::memcpy(lhs, XtX, sizeof(T) * k1 * k1);
      for (int j = 0; j < n; j++) {
        const T c1 = s.conf[j] - (T)1.0;
        const T* col = &s.Xn[(size_t)j * k1];
        for (int b = 0; b < k1; b++) {
          const T f = col[b] * c1;
          for (int a = 0; a < k1; a++) lhs[(size_t)b * k1 + a] += col[a] * f;
        }
      }
      for (int r = 0; r < k1; r++) s.x[r] = rhs_init[r];
      for (int j = 0; j < n; j++) {
        const T w = s.conf[j] - xbn[j] * (s.conf[j] - (T)1.0);                      // :226
        const T* col = &s.Xn[(size_t)j * k1];
        for (int r = 0; r < k1; r++) s.x[r] += col[r] * w;
      }
      if (solver == SEQ_COORDINATE_WISE_NNLS) {
        std::vector<T> rhs(s.x.begin(), s.x.begin() + k1);
        std::memcpy(s.x.data(), init.data(), sizeof(T) * k1);
        c_nnls<T>(lhs, rhs.data(), s.x.data(), k1, SCD_MAX_ITER, SCD_TOL, keep);
      } else if (!solve_sympd(lhs, s.x.data(), k1, keep)) {
        bad += 1;
      }
      std::memcpy(y + ooff, s.x.data(), sizeof(T) * k1);
      T l = 0;
      if (n > 0) {
        gemv_t(s.Xn.data(), s.x.data(), s.t.data(), k1, n);
        for (int j = 0; j < n; j++) {
          const T d = ((T)1.0 - gbias) - s.t[j] - xbn[j];
          l += d * d * s.conf[j];
        }
      }
      loss += l + lambda * dot(s.x.data(), s.x.data(), k1);
    }
  }
  if (lambda > 0) {
    const int r0 = is_x_bias_last_row ? 1 : 0, r1 = is_x_bias_last_row ? k : k - 1;
    double tot = 0;
    for (int j = 0; j < n_rows; j++) {
      const T* col = X + (size_t)j * k;
      T a = 0;
      for (int r = r0; r < r1; r++) a += col[r] * col[r];
      tot += (double)a;
    }
    loss += lambda * tot;
  }
  if (status) *status = bad;
  return loss / (double)nnz;
}

// als_implicit<T> with a global bias, no user/item biases (inst/include/wrmf_implicit.hpp:
// 108-112: global_bias_base = -global_bias * rowSums(X); :155-157 rhs_init = global_bias_base; :178 every column is
// solved; Cholesky / NNLS :228-229 rhs = X_nnz c + rhs_init; conjugate gradient :203 cg_solver_implicit_global_bias from
// the warm start; :262-264 loss against 1 - global_bias).  base_out (k entries) may be null.
template <class T>
double als_implicit_global(int n_rows, int n_cols, const int32_t* col_ptrs, const int32_t* row_indices,
                           const double* values, const T* X, T* Y, const T* XtX, int k, double lambda, int n_threads,
                           unsigned solver, double global_bias_in, T* base_out, int* status, unsigned cg_steps = 3) {
  double loss = 0;
  int bad = 0;
  const T gbias = global_bias_in < std::sqrt((double)std::numeric_limits<T>::epsilon()) ? (T)0 : (T)global_bias_in;
  const size_t nnz = (size_t)col_ptrs[n_cols];
  std::vector<T> base(k, (T)0);
  for (int j = 0; j < n_rows; j++)
    for (int r = 0; r < k; r++) base[r] += X[(size_t)j * k + r];
  for (int r = 0; r < k; r++) base[r] *= -gbias;
  if (base_out) std::memcpy(base_out, base.data(), sizeof(T) * k);
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
  {
    Scratch<T> s;
    std::vector<T> keep;
#pragma omp for schedule(dynamic) reduction(+ : loss) reduction(+ : bad)
    for (int i = 0; i < n_cols; i++) {
      const int p1 = col_ptrs[i], p2 = col_ptrs[i + 1];
      T* y = Y + (size_t)i * k;
      const int n = p2 - p1;
      if (!(gbias != (T)0 || n > 0)) {   // :178, :272-283
        for (int r = 0; r < k; r++) y[r] = 0;
        continue;
      }
      s.ensure(k, n > 0 ? n : 1);
      for (int j = 0; j < n; j++) {
        s.conf[j] = (T)values[p1 + j];
        std::memcpy(&s.Xn[(size_t)j * k], X + (size_t)row_indices[p1 + j] * k, sizeof(T) * k);
      }
      std::vector<T> init(y, y + k);
      if (solver == CONJUGATE_GRADIENT) {   // :199-204
        std::memcpy(s.x.data(), init.data(), sizeof(T) * k);
        if (gbias != (T)0) cg_solver_implicit_global_bias<T>(s, k, n, XtX, cg_steps, base.data(), gbias);
        else cg_solver_implicit<T>(s, k, n, XtX, cg_steps);
        std::memcpy(y, s.x.data(), sizeof(T) * k);
        T lc = 0;
        if (n > 0) {
          gemv_t(s.Xn.data(), s.x.data(), s.t.data(), k, n);
          for (int j = 0; j < n; j++) {
            const T d = ((T)1.0 - gbias) - s.t[j];
            lc += d * d * s.conf[j];
          }
        }
        loss += lc + lambda * dot(s.x.data(), s.x.data(), k);
        continue;
      }
      T* lhs = s.lhs.data();
      std::memcpy(lhs, XtX, sizeof(T) * k * k);
      for (int j = 0; j < n; j++) {
        const T c1 = s.conf[j] - (T)1.0;
        const T* col = &s.Xn[(size_t)j * k];
        for (int b = 0; b < k; b++) {
          const T f = col[b] * c1;
          for (int a = 0; a < k; a++) lhs[(size_t)b * k + a] += col[a] * f;
        }
      }
      for (int r = 0; r < k; r++) s.x[r] = gbias != (T)0 ? base[r] : (T)0;
      for (int j = 0; j < n; j++) {
        const T* col = &s.Xn[(size_t)j * k];
        for (int r = 0; r < k; r++) s.x[r] += col[r] * s.conf[j];
      }
      if (solver == SEQ_COORDINATE_WISE_NNLS) {
        std::vector<T> rhs(s.x.begin(), s.x.begin() + k);
        std::memcpy(s.x.data(), init.data(), sizeof(T) * k);
        c_nnls<T>(lhs, rhs.data(), s.x.data(), k, SCD_MAX_ITER, SCD_TOL, keep);
      } else if (!solve_sympd(lhs, s.x.data(), k, keep)) {
        bad += 1;
      }
      std::memcpy(y, s.x.data(), sizeof(T) * k);
      T l = 0;
      if (n > 0) {
        gemv_t(s.Xn.data(), s.x.data(), s.t.data(), k, n);
        for (int j = 0; j < n; j++) {
          const T d = ((T)1.0 - gbias) - s.t[j];
          l += d * d * s.conf[j];
        }
      }
      loss += l + lambda * dot(s.x.data(), s.x.data(), k);
    }
  }
  if (lambda > 0) {
    double tot = 0;
    for (size_t e = 0; e < (size_t)n_rows * k; e++) tot += (double)(X[e] * X[e]);
    loss += lambda * tot;
  }
  if (status) *status = bad;
  return loss / (double)nnz;
}

// inst/include/wrmf_utils.hpp:86-165 -- initialize_biases_implicit (no global bias here: calculate_global_bias =
// FALSE is what the R driver passes unless with_global_bias, which the device path does not take for implicit
// feedback).  csc = users x items by item column, csr = the same by user column.
template <class T>
double initialize_biases_implicit(int n_items, const int32_t* csc_p, const int32_t* csc_i, const double* csc_x,
                                  int n_users, const int32_t* csr_p, const int32_t* csr_i, const double* csr_x,
                                  T* user_bias, T* item_bias, T lambda, int non_negative, int calculate_global_bias = 0) {
  double global_bias = 0;
  if (calculate_global_bias) {                                                          // :90-93
    long double sm = 0;
    for (int ix = 0; ix < csr_p[n_users]; ix++) sm += (long double)csr_x[ix];
    global_bias = (double)(sm / (sm + (long double)n_items * (long double)n_users - (long double)csr_p[n_users]));
  }
  if (non_negative) global_bias = std::fmax(0., global_bias);
  std::vector<double> user_means(n_users), item_means(n_items), user_adj(n_users, 0.0), item_adj(n_items, 0.0);
  for (int r = 0; r < n_users; r++) {                                                  // :101-112
    const int cnt = csr_p[r + 1] - csr_p[r];
    if (cnt > 0) {
      for (int ix = csr_p[r]; ix < csr_p[r + 1]; ix++) user_adj[r] += csr_x[ix];
      user_means[r] = user_adj[r] / (user_adj[r] + (double)(n_items - cnt));
      user_adj[r] += (double)(n_items - cnt);
      user_adj[r] /= user_adj[r] + lambda;
    } else {
      user_means[r] = 0;
      user_adj[r] = (double)n_items / ((double)n_items + lambda);
    }
  }
  for (int c = 0; c < n_items; c++) {                                                  // :113-124
    const int cnt = csc_p[c + 1] - csc_p[c];
    if (cnt > 0) {
      for (int ix = csc_p[c]; ix < csc_p[c + 1]; ix++) item_adj[c] += csc_x[ix];
      item_means[c] = item_adj[c] / (item_adj[c] + (double)(n_users - cnt));
      item_adj[c] += (double)(n_users - cnt);
      item_adj[c] /= item_adj[c] + lambda;
    } else {
      item_means[c] = 0;
      item_adj[c] = (double)n_users / ((double)n_users + lambda);
    }
  }
  for (int iter = 0; iter < 5; iter++) {                                               // :130-162
    double bias_mean = 0;
    if (iter > 0)
      for (int r = 0; r < n_users; r++) bias_mean += (user_bias[r] - bias_mean) / (T)(r + 1);
    for (int c = 0; c < n_items; c++) {
      double wsum = n_users, bias_this = bias_mean;
      for (int ix = csc_p[c]; ix < csc_p[c + 1]; ix++) {
        wsum += csc_x[ix] - 1;
        bias_this += ((csc_x[ix] - 1) * (user_bias[csc_i[ix]] - bias_this)) / wsum;
      }
      item_bias[c] = (T)((item_means[c] - bias_this - global_bias) * item_adj[c]);
    }
    if (non_negative)
      for (int c = 0; c < n_items; c++) item_bias[c] = std::fmax((T)0, item_bias[c]);
    bias_mean = 0;
    for (int c = 0; c < n_items; c++) bias_mean += (item_bias[c] - bias_mean) / (T)(c + 1);
    for (int r = 0; r < n_users; r++) {
      double wsum = n_items, bias_this = bias_mean;
      for (int ix = csr_p[r]; ix < csr_p[r + 1]; ix++) {
        wsum += csr_x[ix] - 1;
        bias_this += ((csr_x[ix] - 1) * (item_bias[csr_i[ix]] - bias_this)) / wsum;
      }
      user_bias[r] = (T)((user_means[r] - bias_this - global_bias) * user_adj[r]);
    }
    if (non_negative)
      for (int r = 0; r < n_users; r++) user_bias[r] = std::fmax((T)0, user_bias[r]);
  }
  return global_bias;
}

// inst/include/wrmf_utils.hpp:32-84 -- initialize_biases_explicit: optional global mean (running mean, :44-45)
// removed from the values of BOTH orientations in place, then five alternating sweeps
//   item_bias[c] = sum_{u in c} (v - user_bias[u]) / (lambda_use + n_c),  user_bias likewise (:56-81).
// csc = users x items by item column, csr = the same matrix by user column (the reference passes c_ui, c_iu).
template <class T>
double initialize_biases_explicit(int n_items, const int32_t* csc_p, const int32_t* csc_i, double* csc_x,
                                  int n_users, const int32_t* csr_p, const int32_t* csr_i, double* csr_x,
                                  T* user_bias, T* item_bias, T lambda, int dynamic_lambda, int non_negative,
                                  int calculate_global_bias) {
  double global_bias = 0;
  const size_t nnz = (size_t)csc_p[n_items];
  if (calculate_global_bias) {
    for (size_t ix = 0; ix < nnz; ix++) global_bias += (csc_x[ix] - global_bias) / (double)(ix + 1);
    for (size_t ix = 0; ix < nnz; ix++) {
      csc_x[ix] -= global_bias;
      csr_x[ix] -= global_bias;
    }
  }
  for (int iter = 0; iter < 5; iter++) {
    for (int c = 0; c < n_items; c++) item_bias[c] = 0;
    for (int c = 0; c < n_items; c++) {
      const T cnt = (T)(csc_p[c + 1] - csc_p[c]);
      const T lambda_use = lambda * (dynamic_lambda ? cnt : (T)1);
      for (int ix = csc_p[c]; ix < csc_p[c + 1]; ix++) item_bias[c] += (T)(csc_x[ix] - user_bias[csc_i[ix]]);
      item_bias[c] /= lambda_use + cnt;
      if (non_negative) item_bias[c] = std::fmax((T)0, item_bias[c]);
    }
    for (int r = 0; r < n_users; r++) user_bias[r] = 0;
    for (int r = 0; r < n_users; r++) {
      const T cnt = (T)(csr_p[r + 1] - csr_p[r]);
      const T lambda_use = lambda * (dynamic_lambda ? cnt : (T)1);
      for (int ix = csr_p[r]; ix < csr_p[r + 1]; ix++) user_bias[r] += (T)(csr_x[ix] - item_bias[csr_i[ix]]);
      user_bias[r] /= lambda_use + cnt;
      if (non_negative) user_bias[r] = std::fmax((T)0, user_bias[r]);
    }
  }
  return global_bias;
}

// R/model_WRMF.R:474-486 (and :347-353): XtX = tcrossprod(X) + fl(diag(lambda)).
// The ridge passes through float::fl(), i.e. lambda is rounded to fp32 even in the double build.
template <class T>
void gramian(const T* X, int k, int64_t n, double lambda, T* out) {
  std::vector<double> acc((size_t)k * k, 0.0);
#pragma omp parallel
  {
    std::vector<T> loc((size_t)k * k, (T)0);
#pragma omp for schedule(static)
    for (int64_t j = 0; j < n; j++) {
      const T* col = X + (size_t)j * k;
      for (int b = 0; b < k; b++) {
        const T f = col[b];
        for (int a = 0; a < k; a++) loc[(size_t)b * k + a] += col[a] * f;
      }
    }
#pragma omp critical
    for (size_t e = 0; e < (size_t)k * k; e++) acc[e] += (double)loc[e];
  }
  const T ridge = (T)(float)lambda;
  for (int b = 0; b < k; b++)
    for (int a = 0; a < k; a++) out[(size_t)b * k + a] = (T)acc[(size_t)b * k + a] + (a == b ? ridge : (T)0);
}

}  // namespace

extern "C" {

double wrmf_oracle_als_implicit_f32(int n_rows, int n_cols, const int32_t* col_ptrs,
                                    const int32_t* row_indices, const double* values,
                                    const float* X, float* Y, const float* XtX, int k,
                                    double lambda, int n_threads, unsigned solver,
                                    unsigned cg_steps, int* status) {
  return als_implicit<float>(n_rows, n_cols, col_ptrs, row_indices, values, X, Y, XtX, k, lambda,
                             n_threads, solver, cg_steps, status);
}
double wrmf_oracle_als_implicit_f64(int n_rows, int n_cols, const int32_t* col_ptrs,
                                    const int32_t* row_indices, const double* values,
                                    const double* X, double* Y, const double* XtX, int k,
                                    double lambda, int n_threads, unsigned solver,
                                    unsigned cg_steps, int* status) {
  return als_implicit<double>(n_rows, n_cols, col_ptrs, row_indices, values, X, Y, XtX, k, lambda,
                              n_threads, solver, cg_steps, status);
}
double wrmf_oracle_als_implicit_bias_f32(int n_rows, int n_cols, const int32_t* col_ptrs, const int32_t* row_indices,
                                         const double* values, const float* X, float* Y, const float* XtX, int k,
                                         double lambda, int n_threads, unsigned solver, int is_x_bias_last_row,
                                         int* status) {
  return als_implicit_biases<float>(n_rows, n_cols, col_ptrs, row_indices, values, X, Y, XtX, k, lambda, n_threads,
                                    solver, is_x_bias_last_row, status);
}
double wrmf_oracle_als_implicit_bias_f64(int n_rows, int n_cols, const int32_t* col_ptrs, const int32_t* row_indices,
                                         const double* values, const double* X, double* Y, const double* XtX, int k,
                                         double lambda, int n_threads, unsigned solver, int is_x_bias_last_row,
                                         int* status) {
  return als_implicit_biases<double>(n_rows, n_cols, col_ptrs, row_indices, values, X, Y, XtX, k, lambda, n_threads,
                                     solver, is_x_bias_last_row, status);
}
double wrmf_oracle_als_implicit_gbias_f32(int n_rows, int n_cols, const int32_t* col_ptrs, const int32_t* row_indices,
                                          const double* values, const float* X, float* Y, const float* XtX, int k,
                                          double lambda, int n_threads, unsigned solver, int with_biases,
                                          int is_x_bias_last_row, double global_bias, float* base_out, int* status,
                                          unsigned cg_steps) {
  if (with_biases)
    return als_implicit_biases<float>(n_rows, n_cols, col_ptrs, row_indices, values, X, Y, XtX, k, lambda, n_threads,
                                      solver, is_x_bias_last_row, status, global_bias);
  return als_implicit_global<float>(n_rows, n_cols, col_ptrs, row_indices, values, X, Y, XtX, k, lambda, n_threads, solver,
                                    global_bias, base_out, status, cg_steps);
}
double wrmf_oracle_als_implicit_gbias_f64(int n_rows, int n_cols, const int32_t* col_ptrs, const int32_t* row_indices,
                                          const double* values, const double* X, double* Y, const double* XtX, int k,
                                          double lambda, int n_threads, unsigned solver, int with_biases,
                                          int is_x_bias_last_row, double global_bias, double* base_out, int* status,
                                          unsigned cg_steps) {
  if (with_biases)
    return als_implicit_biases<double>(n_rows, n_cols, col_ptrs, row_indices, values, X, Y, XtX, k, lambda, n_threads,
                                       solver, is_x_bias_last_row, status, global_bias);
  return als_implicit_global<double>(n_rows, n_cols, col_ptrs, row_indices, values, X, Y, XtX, k, lambda, n_threads,
                                     solver, global_bias, base_out, status, cg_steps);
}
double wrmf_oracle_init_biases_implicit_f32(int n_items, const int32_t* csc_p, const int32_t* csc_i, const double* csc_x,
                                            int n_users, const int32_t* csr_p, const int32_t* csr_i, const double* csr_x,
                                            float* user_bias, float* item_bias, double lambda, int non_negative,
                                            int calculate_global_bias) {
  return initialize_biases_implicit<float>(n_items, csc_p, csc_i, csc_x, n_users, csr_p, csr_i, csr_x, user_bias,
                                           item_bias, (float)lambda, non_negative, calculate_global_bias);
}
double wrmf_oracle_init_biases_implicit_f64(int n_items, const int32_t* csc_p, const int32_t* csc_i, const double* csc_x,
                                            int n_users, const int32_t* csr_p, const int32_t* csr_i, const double* csr_x,
                                            double* user_bias, double* item_bias, double lambda, int non_negative,
                                            int calculate_global_bias) {
  return initialize_biases_implicit<double>(n_items, csc_p, csc_i, csc_x, n_users, csr_p, csr_i, csr_x, user_bias,
                                            item_bias, lambda, non_negative, calculate_global_bias);
}
double wrmf_oracle_als_explicit_bias_f32(int n_rows, int n_cols, const int32_t* col_ptrs,
                                         const int32_t* row_indices, const double* values, const float* X, float* Y,
                                         const float* cnt_X, int k, double lambda, int n_threads, unsigned solver,
                                         unsigned cg_steps, int dynamic_lambda, int is_x_bias_last_row, int* status) {
  return als_explicit_biases<float>(n_rows, n_cols, col_ptrs, row_indices, values, X, Y, cnt_X, k, lambda, n_threads,
                                    solver, cg_steps, dynamic_lambda, is_x_bias_last_row, status);
}
double wrmf_oracle_als_explicit_bias_f64(int n_rows, int n_cols, const int32_t* col_ptrs,
                                         const int32_t* row_indices, const double* values, const double* X, double* Y,
                                         const double* cnt_X, int k, double lambda, int n_threads, unsigned solver,
                                         unsigned cg_steps, int dynamic_lambda, int is_x_bias_last_row, int* status) {
  return als_explicit_biases<double>(n_rows, n_cols, col_ptrs, row_indices, values, X, Y, cnt_X, k, lambda, n_threads,
                                     solver, cg_steps, dynamic_lambda, is_x_bias_last_row, status);
}
double wrmf_oracle_init_biases_explicit_f32(int n_items, const int32_t* csc_p, const int32_t* csc_i, double* csc_x,
                                            int n_users, const int32_t* csr_p, const int32_t* csr_i, double* csr_x,
                                            float* user_bias, float* item_bias, double lambda, int dynamic_lambda,
                                            int non_negative, int calculate_global_bias) {
  return initialize_biases_explicit<float>(n_items, csc_p, csc_i, csc_x, n_users, csr_p, csr_i, csr_x, user_bias,
                                           item_bias, (float)lambda, dynamic_lambda, non_negative, calculate_global_bias);
}
double wrmf_oracle_init_biases_explicit_f64(int n_items, const int32_t* csc_p, const int32_t* csc_i, double* csc_x,
                                            int n_users, const int32_t* csr_p, const int32_t* csr_i, double* csr_x,
                                            double* user_bias, double* item_bias, double lambda, int dynamic_lambda,
                                            int non_negative, int calculate_global_bias) {
  return initialize_biases_explicit<double>(n_items, csc_p, csc_i, csc_x, n_users, csr_p, csr_i, csr_x, user_bias,
                                            item_bias, lambda, dynamic_lambda, non_negative, calculate_global_bias);
}
double wrmf_oracle_als_explicit_f32(int n_rows, int n_cols, const int32_t* col_ptrs,
                                    const int32_t* row_indices, const double* values,
                                    const float* X, float* Y, const float* cnt_X, int k,
                                    double lambda, int n_threads, unsigned solver,
                                    unsigned cg_steps, int dynamic_lambda, int* status) {
  return als_explicit<float>(n_rows, n_cols, col_ptrs, row_indices, values, X, Y, cnt_X, k,
                             lambda, n_threads, solver, cg_steps, dynamic_lambda, status);
}
double wrmf_oracle_als_explicit_f64(int n_rows, int n_cols, const int32_t* col_ptrs,
                                    const int32_t* row_indices, const double* values,
                                    const double* X, double* Y, const double* cnt_X, int k,
                                    double lambda, int n_threads, unsigned solver,
                                    unsigned cg_steps, int dynamic_lambda, int* status) {
  return als_explicit<double>(n_rows, n_cols, col_ptrs, row_indices, values, X, Y, cnt_X, k,
                              lambda, n_threads, solver, cg_steps, dynamic_lambda, status);
}
void wrmf_oracle_gramian_f32(const float* X, int k, int64_t n, double lambda, float* out) {
  gramian<float>(X, k, n, lambda, out);
}
void wrmf_oracle_gramian_f64(const double* X, int k, int64_t n, double lambda, double* out) {
  gramian<double>(X, k, n, lambda, out);
}
int wrmf_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

}  // extern "C"
