// User/item-bias support for explicit feedback (gfx950): the pieces around the unchanged solve kernels.
//
// als_explicit<T> with_biases (inst/include/wrmf_explicit.hpp:41-64,86-91,113-127): the model carries two extra
// coordinates -- a row of ones that multiplies the other side's bias and the own bias.  Per row the reference
// drops the x_bias row from X_nnz, subtracts the fixed side's biases from the ratings and solves a (rank-1)-system
// exactly as in the no-bias branch.  That is done here by re-packing, not by a second set of solve kernels:
//     X' = X without its x_bias row (strided copy), r'_j = r_j - x_bias[idx_j] (bias_shift_values_kernel),
//     y'_0 = the warm start the reference uses (drop_row(init, !is_x_bias_last_row), :90), solve with rank-1,
//     copy y' back to the head / tail of Y.col(i).
// initialize_biases_explicit (inst/include/wrmf_utils.hpp:32-84): optional global mean removed from the values of
// both orientations, then five alternating sweeps  bias[c] = sum_{e in c}(v_e - other_bias[idx_e]) / (lambda_use + n_c),
// one wave per column, sums in double.
#include <algorithm>

#include "wrmf_internal.h"
#include "wrmf_f64.h"
#include "wrmf_device.h"

namespace rsparse_hip {
namespace {

using namespace dev;

__global__ __launch_bounds__(256) void bias_shift_values_kernel(const float* __restrict__ vals,
                                                                const int32_t* __restrict__ row_idx,
                                                                const float* __restrict__ X, int k, int bias_row,
                                                                int64_t nnz, float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += stride)
    out[e] = vals[e] - X[(size_t)row_idx[e] * k + bias_row];
}

// T = float (the device-resident fp32 layer) or double (wrmf_f64.h: the *_double entry points)
template <class T>
__global__ __launch_bounds__(256) void bias_sweep_kernel(const int32_t* __restrict__ p, const int32_t* __restrict__ i,
                                                         const T* __restrict__ x, const T* __restrict__ other,
                                                         int n_cols, T lambda, int dynamic_lambda, int non_negative,
                                                         T* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = (int)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6);
  const int n_waves = (int)((gridDim.x * (size_t)blockDim.x) >> 6);
  for (int c = wave; c < n_cols; c += n_waves) {
    const int p1 = p[c], p2 = p[c + 1];
    double s = 0.0;
    for (int e = p1 + lane; e < p2; e += 64) s += (double)x[e] - (double)other[i[e]];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if (lane == 0) {
      const T cnt = (T)(p2 - p1);
      const T lambda_use = lambda * (dynamic_lambda ? cnt : (T)1);
      T b = (T)s / (lambda_use + cnt);   // an empty column with lambda = 0 gives 0/0 = NaN, as in the reference
      if (non_negative) b = b > (T)0 ? b : (T)0;   // std::fmax(0, b) (NaN -> 0, as fmax defines it)
      out[c] = b;
    }
  }
}

template <class T>
__global__ __launch_bounds__(256) void values_sum_kernel(const T* __restrict__ x, int64_t n, double* __restrict__ partials) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double s = 0.0;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) s += (double)x[e];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
  if (lane == 0) partials[(size_t)blockIdx.x * 4 + wv] = s;
}

template <class T>
__global__ __launch_bounds__(256) void values_add_kernel(T* __restrict__ x, int64_t n, const double* __restrict__ sum,
                                                         double inv_count) {
  const T shift = (T)(sum[0] * inv_count);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) x[e] -= shift;
}

// implicit feedback with biases (inst/include/wrmf_implicit.hpp:226,256-270): per non-zero, the coefficient of x_j in
// the right-hand side, c - x_b (c - 1), and the target of the loss term, 1 - x_b
__global__ __launch_bounds__(256) void bias_implicit_terms_kernel(const float* __restrict__ vals,
                                                                  const int32_t* __restrict__ row_idx,
                                                                  const float* __restrict__ X, int k, int bias_row,
                                                                  int64_t nnz, float global_bias,
                                                                  float* __restrict__ rhs_vals,
                                                                  float* __restrict__ loss_tgt) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += stride) {
    const float c = vals[e], xb = X[(size_t)row_idx[e] * k + bias_row];
    rhs_vals[e] = c - xb * (c - 1.f);
    loss_tgt[e] = (1.f - global_bias) - xb;   // wrmf_implicit.hpp:265-270
  }
}

// rhs_init = -drop_row(X) * x_biases (wrmf_implicit.hpp:142-147): partial[b][t] = -sum over the block's entities of
// X[e][off + t] * X[e][bias_row]; fixed-order second stage -> deterministic
constexpr int kRhsInitBlocks = 256;
constexpr int kRhsInitW = 256;   // >= the largest system order (RSPARSE_HIP_MAX_RANK)
__global__ __launch_bounds__(kRhsInitW) void bias_rhs_init_partial_kernel(const float* __restrict__ X, int k, int off, int k1,
                                                                    int bias_row, float global_bias, int n,
                                                                    float* __restrict__ partial) {
  const int t = threadIdx.x;
  const int per = (n + gridDim.x - 1) / gridDim.x;
  const int e0 = blockIdx.x * per, e1 = min(n, e0 + per);
  float s = 0.f;
  if (t < k1)
    for (int e = e0; e < e1; e++)   // rhs_init = -X' (x_b + global_bias)  (wrmf_implicit.hpp:146-153; :110-112 without biases)
      s = fmaf(-X[(size_t)e * k + off + t], (bias_row >= 0 ? X[(size_t)e * k + bias_row] : 0.f) + global_bias, s);
  partial[(size_t)blockIdx.x * kRhsInitW + t] = s;
}
__global__ __launch_bounds__(kRhsInitW) void bias_rhs_init_reduce_kernel(const float* __restrict__ partial, int blocks,
                                                                   float* __restrict__ out) {
  const int t = threadIdx.x;
  float s = 0.f;
  for (int b = 0; b < blocks; b++) s += partial[(size_t)b * kRhsInitW + t];
  out[t] = s;
}

// initialize_biases_implicit (inst/include/wrmf_utils.hpp:86-165), one thread per column (the weighted running mean
// over a column's entries is a serial recurrence).  Stage "prep": means[c], adj[c] (:101-124).
template <class T>
__global__ __launch_bounds__(256) void bias_implicit_prep_kernel(const int32_t* __restrict__ p, const T* __restrict__ x,
                                                                 int n_cols, int n_other, double lambda,
                                                                 double* __restrict__ means, double* __restrict__ adj) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cols) return;
  const int p1 = p[c], p2 = p[c + 1];
  if (p2 > p1) {
    double a = 0.0;
    for (int e = p1; e < p2; e++) a += (double)x[e];
    const double rest = (double)(n_other - (p2 - p1));
    means[c] = a / (a + rest);
    a += rest;
    adj[c] = a / (a + lambda);
  } else {
    means[c] = 0.0;
    adj[c] = (double)n_other / ((double)n_other + lambda);
  }
}
// one sweep (:136-143 / :152-159): bias[c] = (means[c] - running weighted mean of the other side's biases) * adj[c]
template <class T>
__global__ __launch_bounds__(256) void bias_implicit_sweep_kernel(const int32_t* __restrict__ p, const int32_t* __restrict__ i,
                                                                  const T* __restrict__ x,
                                                                  const T* __restrict__ other, int n_cols, int n_other,
                                                                  const double* __restrict__ other_sum,
                                                                  const double* __restrict__ means,
                                                                  const double* __restrict__ adj, int non_negative,
                                                                  double global_bias, T* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cols) return;
  double wsum = (double)n_other;
  double bias_this = other_sum ? other_sum[0] / (double)n_other : 0.0;   // bias_mean (:131-135,148-150)
  for (int e = p[c]; e < p[c + 1]; e++) {
    const double w = (double)x[e] - 1.0;
    wsum += w;
    bias_this += (w * ((double)other[i[e]] - bias_this)) / wsum;
  }
  T b = (T)((means[c] - bias_this - global_bias) * adj[c]);   // wrmf_utils.hpp:142,157
  if (non_negative) b = b > (T)0 ? b : (T)0;
  out[c] = b;
}

}  // namespace

hipError_t launch_bias_implicit_terms(const float* vals, const int32_t* row_idx, const float* X, int k, int bias_row,
                                      int64_t nnz, float global_bias, float* rhs_vals, float* loss_tgt, hipStream_t s) {
  if (nnz <= 0) return hipSuccess;
  hipLaunchKernelGGL(bias_implicit_terms_kernel, dim3(2048), dim3(256), 0, s, vals, row_idx, X, k, bias_row, nnz,
                     global_bias, rhs_vals, loss_tgt);
  return hipGetLastError();
}

// out[0..256): rhs_init (entries >= k1 are zero); scratch: kRhsInitBlocks * 256 floats
hipError_t launch_bias_rhs_init(const float* X, int k, int off, int k1, int bias_row, float global_bias, int n,
                                float* scratch, float* out, hipStream_t s) {
  hipLaunchKernelGGL(bias_rhs_init_partial_kernel, dim3(kRhsInitBlocks), dim3(kRhsInitW), 0, s, X, k, off, k1, bias_row,
                     global_bias, n, scratch);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(bias_rhs_init_reduce_kernel, dim3(1), dim3(kRhsInitW), 0, s, scratch, kRhsInitBlocks, out);
  return hipGetLastError();
}
size_t bias_rhs_init_scratch_floats() { return (size_t)kRhsInitBlocks * kRhsInitW + kRhsInitW; }

// ---- the re-packed operands at a rank that is a multiple of 4 ----
// The solves of a biased half-iteration run at rank - 1 (the x-bias row dropped): 65 for the reference's rank = 64 + 2, an odd
// number for every even user rank -- which the register-resident kernels do not take (their vectors are 16-byte pieces; the
// LDS-tile fallback is an order of magnitude slower).  Coordinates of zeros change nothing: X' has no component there, the
// warm start is 0 there, the system gets lambda (explicit) or 1 (the padded Gramian's diagonal) there against a right-hand side
// of 0, so y stays 0 there under every solver, and the loss sums see zeros.  So the copies are padded to the next multiple of 4.
namespace {
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ src, int src_stride, int src_off, int k1,
                                                       int k1p, int64_t n, float* __restrict__ dst) {
  const int64_t total = n * k1p;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / k1p;
    const int c = (int)(e - r * k1p);
    dst[e] = c < k1 ? src[r * src_stride + src_off + c] : 0.f;
  }
}
__global__ __launch_bounds__(256) void pad_gramian_kernel(const float* __restrict__ G, int k1, int k1p, float* __restrict__ Gp) {
  for (int e = threadIdx.x; e < k1p * k1p; e += 256) {
    const int r = e / k1p, c = e % k1p;
    Gp[e] = (r < k1 && c < k1) ? G[r * k1 + c] : (r == c ? 1.f : 0.f);
  }
}
}  // namespace

hipError_t launch_pad_rows(const float* src, int src_stride, int src_off, int k1, int k1p, int64_t n, float* dst, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const int64_t total = n * k1p;
  const int grid = (int)std::min<int64_t>((total + 255) / 256, 65536);
  hipLaunchKernelGGL(pad_rows_kernel, dim3(grid), dim3(256), 0, s, src, src_stride, src_off, k1, k1p, n, dst);
  return hipGetLastError();
}
hipError_t launch_pad_gramian(const float* G, int k1, int k1p, float* Gp, hipStream_t s) {
  hipLaunchKernelGGL(pad_gramian_kernel, dim3(1), dim3(256), 0, s, G, k1, k1p, Gp);
  return hipGetLastError();
}

namespace {
template <class T>
hipError_t bias_implicit_prep_t(const int32_t* p, const T* x, int n_cols, int n_other, double lambda, double* means,
                                double* adj, hipStream_t s) {
  if (n_cols <= 0) return hipSuccess;
  hipLaunchKernelGGL(bias_implicit_prep_kernel<T>, dim3((n_cols + 255) / 256), dim3(256), 0, s, p, x, n_cols, n_other,
                     lambda, means, adj);
  return hipGetLastError();
}
template <class T>
hipError_t bias_implicit_sweep_t(const int32_t* p, const int32_t* i, const T* x, const T* other, int n_cols, int n_other,
                                 const double* other_sum, const double* means, const double* adj, int non_negative,
                                 double global_bias, T* out, hipStream_t s) {
  if (n_cols <= 0) return hipSuccess;
  hipLaunchKernelGGL(bias_implicit_sweep_kernel<T>, dim3((n_cols + 255) / 256), dim3(256), 0, s, p, i, x, other, n_cols,
                     n_other, other_sum, means, adj, non_negative, global_bias, out);
  return hipGetLastError();
}
}  // namespace

hipError_t launch_bias_implicit_prep(const int32_t* p, const float* x, int n_cols, int n_other, double lambda,
                                     double* means, double* adj, hipStream_t s) {
  return bias_implicit_prep_t<float>(p, x, n_cols, n_other, lambda, means, adj, s);
}
hipError_t launch_bias_implicit_prep(const int32_t* p, const double* x, int n_cols, int n_other, double lambda,
                                     double* means, double* adj, hipStream_t s) {
  return bias_implicit_prep_t<double>(p, x, n_cols, n_other, lambda, means, adj, s);
}
hipError_t launch_bias_implicit_sweep(const int32_t* p, const int32_t* i, const float* x, const float* other, int n_cols,
                                      int n_other, const double* other_sum, const double* means, const double* adj,
                                      int non_negative, double global_bias, float* out, hipStream_t s) {
  return bias_implicit_sweep_t<float>(p, i, x, other, n_cols, n_other, other_sum, means, adj, non_negative, global_bias, out, s);
}
hipError_t launch_bias_implicit_sweep(const int32_t* p, const int32_t* i, const double* x, const double* other, int n_cols,
                                      int n_other, const double* other_sum, const double* means, const double* adj,
                                      int non_negative, double global_bias, double* out, hipStream_t s) {
  return bias_implicit_sweep_t<double>(p, i, x, other, n_cols, n_other, other_sum, means, adj, non_negative, global_bias, out, s);
}

hipError_t launch_bias_shift_values(const float* vals, const int32_t* row_idx, const float* X, int k, int bias_row,
                                    int64_t nnz, float* out, hipStream_t s) {
  if (nnz <= 0) return hipSuccess;
  hipLaunchKernelGGL(bias_shift_values_kernel, dim3(2048), dim3(256), 0, s, vals, row_idx, X, k, bias_row, nnz, out);
  return hipGetLastError();
}

hipError_t launch_bias_sweep(const int32_t* p, const int32_t* i, const float* x, const float* other, int n_cols,
                             float lambda, int dynamic_lambda, int non_negative, float* out, hipStream_t s) {
  if (n_cols <= 0) return hipSuccess;
  hipLaunchKernelGGL(bias_sweep_kernel<float>, dim3(2048), dim3(256), 0, s, p, i, x, other, n_cols, lambda, dynamic_lambda,
                     non_negative, out);
  return hipGetLastError();
}
hipError_t launch_bias_sweep(const int32_t* p, const int32_t* i, const double* x, const double* other, int n_cols,
                             double lambda, int dynamic_lambda, int non_negative, double* out, hipStream_t s) {
  if (n_cols <= 0) return hipSuccess;
  hipLaunchKernelGGL(bias_sweep_kernel<double>, dim3(2048), dim3(256), 0, s, p, i, x, other, n_cols, lambda, dynamic_lambda,
                     non_negative, out);
  return hipGetLastError();
}

// sum of x[0..n) in double -> out[0] (partials: >= 1024 doubles)
hipError_t launch_values_sum(const float* x, int64_t n, double* partials, double* out, hipStream_t s) {
  hipLaunchKernelGGL(values_sum_kernel<float>, dim3(256), dim3(256), 0, s, x, n, partials);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return launch_sum_partials(partials, 1024, out, s);
}
hipError_t launch_values_sum(const double* x, int64_t n, double* partials, double* out, hipStream_t s) {
  hipLaunchKernelGGL(values_sum_kernel<double>, dim3(256), dim3(256), 0, s, x, n, partials);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return launch_sum_partials(partials, 1024, out, s);
}

// x[e] -= sum[0] * inv_count
hipError_t launch_values_subtract_mean(float* x, int64_t n, const double* sum, double inv_count, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(values_add_kernel<float>, dim3(2048), dim3(256), 0, s, x, n, sum, inv_count);
  return hipGetLastError();
}
hipError_t launch_values_subtract_mean(double* x, int64_t n, const double* sum, double inv_count, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(values_add_kernel<double>, dim3(2048), dim3(256), 0, s, x, n, sum, inv_count);
  return hipGetLastError();
}

}  // namespace rsparse_hip
