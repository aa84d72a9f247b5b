// fp64 device path (wrmf_f64.hip): what the reference's `*_double` entry points compute -- als_implicit<double> /
// als_explicit<double> (src/wrmf_implicit.cpp:5-14, src/wrmf_explicit.cpp:5-14; R/model_WRMF.R:82: precision = "double"
// is the constructor's default) -- in double on the device.  Internal interface between wrmf_f64_capi.cpp and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace rsparse_hip {

struct F64Args {
  const int32_t* col_ptrs;
  const int32_t* row_idx;
  const double* vals;
  const double* X;      // k x n_rows, ld = k
  double* Y;            // k x n_cols, ld = k
  const double* XtX;    // k1 x k1 column-major, ridge included (implicit) or nullptr (explicit)
  int n_cols;
  int k;                // leading dimension of X and Y: the rank, including the two bias coordinates when there are biases
  int k1;               // order of the per-row system: k, or k - 1 with user/item biases
  // user/item biases (wrmf_explicit.hpp:41-64,86-91,113-127; wrmf_implicit.hpp:114-154,186-252): X_nnz = rows
  // [xoff, xoff + k1) of X's columns, x_biases = row xb (-1: none), warm start = Y[ioff ..], result -> Y[ooff ..]
  int xoff, xb, ioff, ooff;
  int implicit;
  int solver;           // wrmf.hpp:16-18
  int cg_steps;
  double lambda;
  int dynamic_lambda;
  double gbias;         // implicit feedback: the global bias, already thresholded (wrmf_implicit.hpp:108-109); 0 = none
  const double* rhs_init;   // k1 doubles added to every right-hand side (global_bias_base / -X'(x_b + g)), or nullptr
  int solve_empty;      // implicit feedback with biases or a global bias: empty columns are solved too (:178)
  double* loss_partials;    // one double per workgroup
  int* fail_counter;    // [2] += rows re-solved by the general solver, [3] += rows singular for it too
  double* m2_scratch;   // NNLS when two matrices do not fit the LDS: per workgroup KP * (KP + 1) doubles
  // conjugate gradient, rows of more than long_min non-zeros ("long rows", wrmf_f64.hip): cut into chunks of chunk_len non-zeros
  // that run as waves of their own, pass by pass -- one wave per row left a half-iteration waiting for its longest row
  int long_min;                 // rows up to this length stay with the wave-per-row kernel
  int chunk_len;
  int n_long, n_chunks;
  const int32_t* long_rows;     // [n_long] the rows, ascending
  const int32_t* long_chunk0;   // [n_long + 1] first chunk of each
  const int32_t* chunk_long;    // [n_chunks] position in long_rows
  const int32_t* chunk_off;     // [n_chunks] first non-zero of the chunk within its row
  double* long_scratch;         // f64_long_scratch_doubles(): the chunks' partial sums, r and p of every long row, its scalars
};
size_t f64_long_scratch_doubles(int k, int n_long, int n_chunks);

constexpr int kF64MaxGrid = 256 * 48;      // workgroups of the per-row kernel (grid-stride over the rows; the small-rank kernel is one wave and little LDS: up to 32 per CU are resident)
constexpr int kF64GramBlocks = 256;        // partial Gramians
size_t f64_gramian_scratch_doubles(int k);
// XtX = X X^T + ridge I (k x k column-major), sumsq (nullable) = trace before the ridge
hipError_t launch_f64_gramian(const double* X, int k, int64_t n, double ridge, double* XtX, double* sumsq, double* scratch,
                              hipStream_t s);
// grid used by launch_f64_als for n_cols rows (= loss partial slots it writes)
int f64_als_grid(int n_cols);
bool f64_needs_m2_scratch(int k1, int solver);
size_t f64_m2_doubles_per_wg(int k1);
hipError_t launch_f64_als(const F64Args& a, hipStream_t s);
// out[t] = - sum_e X[off + t, e] * ((bias_row >= 0 ? X[bias_row, e] : 0) + global_bias), t < k1; scratch: 256 * 128 doubles
hipError_t launch_f64_rhs_init(const double* X, int k, int off, int k1, int bias_row, double global_bias, int n,
                               double* scratch, double* out, hipStream_t s);
// out[0] = sum_j w_j |X[:, j]|^2 (w = nullptr: 1); partials: >= 1024 doubles
hipError_t launch_f64_weighted_sumsq(const double* X, int k, int64_t n, const double* w, double* out, double* partials,
                                     hipStream_t s);

// explicit feedback with biases, conjugate gradient: the operands re-packed for the wave-per-row kernels (wrmf_f64.hip)
hipError_t launch_f64_pack_rows(const double* src, int ld, int off, int k1, int64_t n, double* dst, hipStream_t s);
hipError_t launch_f64_unpack_rows(const double* src, int k1, int64_t n, int ld, int off, double* dst, hipStream_t s);
hipError_t launch_f64_shift_values(const double* vals, const int32_t* idx, const double* X, int ld, int xb, int64_t nnz,
                                   double* out, hipStream_t s);

// double overloads of the bias-initialisation pieces (wrmf_bias.hip)
hipError_t launch_bias_sweep(const int32_t* p, const int32_t* i, const double* x, const double* other, int n_cols,
                             double lambda, int dynamic_lambda, int non_negative, double* out, hipStream_t s);
hipError_t launch_values_sum(const double* x, int64_t n, double* partials, double* out, hipStream_t s);
hipError_t launch_values_subtract_mean(double* x, int64_t n, const double* sum, double inv_count, hipStream_t s);
hipError_t launch_bias_implicit_prep(const int32_t* p, const double* x, int n_cols, int n_other, double lambda,
                                     double* means, double* adj, hipStream_t s);
hipError_t launch_bias_implicit_sweep(const int32_t* p, const int32_t* i, const double* x, const double* other, int n_cols,
                                      int n_other, const double* other_sum, const double* means, const double* adj,
                                      int non_negative, double global_bias, double* out, hipStream_t s);

}  // namespace rsparse_hip
