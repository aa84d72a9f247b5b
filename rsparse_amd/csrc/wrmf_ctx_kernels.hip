// The three device helpers of the multi-GPU context (wrmf_ctx.cpp): what `ShardedALS.gramian` of rsparse_amd/engine.py does with
// torch ops -- the ranks' Gramian partials go through ONE exchange as doubles (k x k partial, sum(F^2), max |F|), every rank sums
// the contributions in rank order (deterministic, identical everywhere) and adds the fp32-rounded ridge
// (fl(diag(lambda)), R/model_WRMF.R:476).
#include "wrmf_internal.h"

namespace rsparse_hip {
namespace {

// red[0 .. kk) += Gpart, red[kk] += *sumsq  (one block of a rank's rows at a time)
__global__ void ctx_accumulate_kernel(const float* __restrict__ Gpart, const double* __restrict__ sumsq, double* __restrict__ red, int kk) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < kk) red[e] += (double)Gpart[e];
  if (e == 0) red[kk] += *sumsq;
}

// red[kk + 1] = max |F| of the rank's rows (it rode along with the partials)
__global__ void ctx_put_absmax_kernel(const float* __restrict__ absmax, double* __restrict__ red, int kk) { red[kk + 1] = (double)*absmax; }

// G = sum over the ranks (rank order) + ridge on the diagonal; scal[0] = sum(F^2); absmax = max over the ranks
__global__ void ctx_reduce_kernel(const double* __restrict__ all, int ws, int k, float ridge, float* __restrict__ G, double* __restrict__ scal0,
                                  float* __restrict__ absmax) {
  const int kk = k * k, n = kk + 2;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < kk) {
    double s = 0.0;
    for (int r = 0; r < ws; r++) s += all[(size_t)r * n + e];
    float g = (float)s;
    if (e / k == e % k) g += ridge;
    G[e] = g;
  }
  if (e == 0) {
    double s = 0.0, m = 0.0;
    for (int r = 0; r < ws; r++) {
      s += all[(size_t)r * n + kk];
      m = fmax(m, all[(size_t)r * n + kk + 1]);
    }
    *scal0 = s;
    *absmax = (float)m;
  }
}

// out[0] = sum of n doubles (the sub-blocks' loss terms)
__global__ void ctx_sum_kernel(const double* __restrict__ v, int n, double* __restrict__ out) {
  double s = 0.0;
  for (int i = 0; i < n; i++) s += v[i];
  *out = s;
}

__global__ void ctx_add_kernel(const double* __restrict__ src, double* __restrict__ dst) { *dst += *src; }

}  // namespace

hipError_t launch_ctx_add(const double* src, double* dst, hipStream_t s) {
  hipLaunchKernelGGL(ctx_add_kernel, dim3(1), dim3(1), 0, s, src, dst);
  return hipGetLastError();
}
hipError_t launch_ctx_accumulate(const float* Gpart, const double* sumsq, double* red, int k, hipStream_t s) {
  const int kk = k * k;
  hipLaunchKernelGGL(ctx_accumulate_kernel, dim3((kk + 255) / 256), dim3(256), 0, s, Gpart, sumsq, red, kk);
  return hipGetLastError();
}
hipError_t launch_ctx_put_absmax(const float* absmax, double* red, int k, hipStream_t s) {
  hipLaunchKernelGGL(ctx_put_absmax_kernel, dim3(1), dim3(1), 0, s, absmax, red, k * k);
  return hipGetLastError();
}
hipError_t launch_ctx_reduce(const double* all, int ws, int k, float ridge, float* G, double* scal0, float* absmax, hipStream_t s) {
  const int kk = k * k;
  hipLaunchKernelGGL(ctx_reduce_kernel, dim3((kk + 255) / 256), dim3(256), 0, s, all, ws, k, ridge, G, scal0, absmax);
  return hipGetLastError();
}
hipError_t launch_ctx_sum(const double* v, int n, double* out, hipStream_t s) {
  hipLaunchKernelGGL(ctx_sum_kernel, dim3(1), dim3(1), 0, s, v, n, out);
  return hipGetLastError();
}

}  // namespace rsparse_hip
