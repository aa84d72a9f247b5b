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
#include "wrmf_internal.h"
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

__global__ __launch_bounds__(256) void bias_sweep_kernel(const int32_t* __restrict__ p, const int32_t* __restrict__ i,
                                                         const float* __restrict__ x, const float* __restrict__ other,
                                                         int n_cols, float lambda, int dynamic_lambda, int non_negative,
                                                         float* __restrict__ out) {
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
      const float cnt = (float)(p2 - p1);
      const float lambda_use = lambda * (dynamic_lambda ? cnt : 1.f);
      float b = (float)s / (lambda_use + cnt);   // an empty column with lambda = 0 gives 0/0 = NaN, as in the reference
      if (non_negative) b = fmaxf(0.f, b);
      out[c] = b;
    }
  }
}

__global__ __launch_bounds__(256) void values_sum_kernel(const float* __restrict__ x, int64_t n, double* __restrict__ partials) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double s = 0.0;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) s += (double)x[e];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
  if (lane == 0) partials[(size_t)blockIdx.x * 4 + wv] = s;
}

__global__ __launch_bounds__(256) void values_add_kernel(float* __restrict__ x, int64_t n, const double* __restrict__ sum,
                                                         double inv_count) {
  const float shift = (float)(sum[0] * inv_count);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) x[e] -= shift;
}

}  // namespace

hipError_t launch_bias_shift_values(const float* vals, const int32_t* row_idx, const float* X, int k, int bias_row,
                                    int64_t nnz, float* out, hipStream_t s) {
  if (nnz <= 0) return hipSuccess;
  hipLaunchKernelGGL(bias_shift_values_kernel, dim3(2048), dim3(256), 0, s, vals, row_idx, X, k, bias_row, nnz, out);
  return hipGetLastError();
}

hipError_t launch_bias_sweep(const int32_t* p, const int32_t* i, const float* x, const float* other, int n_cols,
                             float lambda, int dynamic_lambda, int non_negative, float* out, hipStream_t s) {
  if (n_cols <= 0) return hipSuccess;
  hipLaunchKernelGGL(bias_sweep_kernel, dim3(2048), dim3(256), 0, s, p, i, x, other, n_cols, lambda, dynamic_lambda,
                     non_negative, out);
  return hipGetLastError();
}

// sum of x[0..n) in double -> out[0] (partials: >= 1024 doubles)
hipError_t launch_values_sum(const float* x, int64_t n, double* partials, double* out, hipStream_t s) {
  hipLaunchKernelGGL(values_sum_kernel, dim3(256), dim3(256), 0, s, x, n, partials);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return launch_sum_partials(partials, 1024, out, s);
}

// x[e] -= sum[0] * inv_count
hipError_t launch_values_subtract_mean(float* x, int64_t n, const double* sum, double inv_count, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(values_add_kernel, dim3(2048), dim3(256), 0, s, x, n, sum, inv_count);
  return hipGetLastError();
}

}  // namespace rsparse_hip
