// General-solver fallback of the exact (Cholesky-branch) solve (gfx950, wave64).
//
// The reference solves  lhs y = rhs  with arma::solve(lhs, rhs, solve_opts::fast + solve_opts::likely_sympd)
// (inst/include/wrmf_implicit.hpp:236; wrmf_explicit.hpp:108 with `fast` only): LAPACK's Cholesky first and, when that
// factorisation fails, a general LU solve with partial pivoting (gesv) behind a warning.  The Cholesky kernels
// (wrmf_chol.hip, the exact solve of wrmf_ne.hip) append every row whose factorisation met a non-positive pivot to a
// list; this kernel re-assembles those rows' systems and solves them by Gaussian elimination with partial pivoting, the
// same elimination order as gesv -- one workgroup per row: such rows are rare (an indefinite system needs a confidence
// below 1, a singular one explicit feedback with lambda = 0 and fewer ratings than factors), so nothing here is tuned.
// A system whose pivot column is exactly zero is counted as unresolved (the reference would go on to a least-squares
// solve or throw); its row keeps the zeros written here.
#include "wrmf_internal.h"
#include "wrmf_device.h"

namespace rsparse_hip {
namespace {

using namespace dev;

template <bool IMPLICIT>
__global__ __launch_bounds__(256) void als_lu_fallback_kernel(AlsArgs a, size_t loss_slot0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int k = a.k, ld = k + 1;
  float* sA = reinterpret_cast<float*>(smem);   // [k][ld]   row-major: sA[i * ld + m] = lhs(i, m)
  float* sB = sA + (size_t)k * ld;              // [k]       right-hand side, then the solution
  float* sVv = sB + k;                          // [8][k]    staged factor vectors
  float* sW = sVv + 8 * k;                      // [8]       their weights in the system matrix
  float* sR = sW + 8;                           // [8]       ... and in the right-hand side
  int* sPiv = reinterpret_cast<int*>(sR + 8);   // [2]       pivot row, "singular" flag
  float* sRed = reinterpret_cast<float*>(sPiv + 2);   // [8]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n_fail = min(a.fail_counter[0], a.fail_cap);
  double wloss = 0.0;
  for (int it = blockIdx.x; it < n_fail; it += gridDim.x) {
    const int row = a.fail_rows[it];
    const int p1 = a.col_ptrs[row], cnt = a.col_ptrs[row + 1] - p1;
    const float lam_use = IMPLICIT ? 0.f : (float)(a.lambda_loss * (a.dynamic_lambda ? (double)(float)cnt : 1.0));
    __syncthreads();
    // lhs = XtX (implicit) or lambda_use I (explicit); rhs = rhs_init (biases / global bias) or 0
    for (int e = tid; e < k * k; e += 256) {
      const int i = e / k, m = e - i * k;
      sA[i * ld + m] = IMPLICIT ? a.XtX[(size_t)i * k + m] : (i == m ? lam_use : 0.f);
    }
    for (int e = tid; e < k; e += 256) sB[e] = a.rhs_init ? a.rhs_init[e] : 0.f;
    // + sum_j w_j x_j x_j^T,  w = c - 1 (implicit) / 1 (explicit);  rhs += sum_j coef_j x_j
    for (int j0 = 0; j0 < cnt; j0 += 8) {
      __syncthreads();
      const int nj = min(8, cnt - j0);
      for (int e = tid; e < nj * k; e += 256) {
        const int j = e / k, t = e - j * k;
        sVv[j * k + t] = a.X[(size_t)a.row_idx[p1 + j0 + j] * k + t];
      }
      if (tid < nj) {
        const float c = a.vals[p1 + j0 + tid];
        sW[tid] = IMPLICIT ? c - 1.f : 1.f;
        sR[tid] = a.rhs_vals ? a.rhs_vals[p1 + j0 + tid] : c;
      }
      __syncthreads();
      for (int e = tid; e < k * k; e += 256) {
        const int i = e / k, m = e - i * k;
        float s = sA[i * ld + m];
        for (int j = 0; j < nj; j++) s = fmaf(sW[j] * sVv[j * k + i], sVv[j * k + m], s);
        sA[i * ld + m] = s;
      }
      for (int e = tid; e < k; e += 256) {
        float s = sB[e];
        for (int j = 0; j < nj; j++) s = fmaf(sR[j], sVv[j * k + e], s);
        sB[e] = s;
      }
    }
    __syncthreads();
    // Gaussian elimination with partial pivoting (the largest |entry| of column c among the rows >= c, first one on ties)
    if (tid == 0) sPiv[1] = 0;
    for (int c = 0; c < k; c++) {
      __syncthreads();
      if (wv == 0) {
        float best = -1.f;
        int bi = c;
        for (int i = c + lane; i < k; i += 64) {
          const float v = fabsf(sA[i * ld + c]);
          if (v > best) { best = v; bi = i; }
        }
        for (int o = 32; o > 0; o >>= 1) {
          const float ob = __shfl_xor(best, o);
          const int oi = __shfl_xor(bi, o);
          if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) {
          sPiv[0] = bi;
          if (!(best > 0.f)) sPiv[1] = 1;   // exactly singular (or NaN): no general solution either
        }
      }
      __syncthreads();
      if (sPiv[1]) break;   // uniform
      const int piv = sPiv[0];
      if (piv != c) {
        for (int m = tid; m < k; m += 256) {
          const float t = sA[c * ld + m];
          sA[c * ld + m] = sA[piv * ld + m];
          sA[piv * ld + m] = t;
        }
        if (tid == 0) {
          const float t = sB[c];
          sB[c] = sB[piv];
          sB[piv] = t;
        }
      }
      __syncthreads();
      const float pinv = 1.f / sA[c * ld + c];
      const float bc = sB[c];
      for (int i = c + 1 + (tid >> 4); i < k; i += 16) {
        const float f = sA[i * ld + c] * pinv;
        for (int m = c + 1 + (tid & 15); m < k; m += 16) sA[i * ld + m] = fmaf(-f, sA[c * ld + m], sA[i * ld + m]);
        if ((tid & 15) == 0) sB[i] = fmaf(-f, bc, sB[i]);
      }
      // (column c below the pivot is left as it is: nothing reads it again)
    }
    __syncthreads();
    const bool singular = sPiv[1] != 0;
    float* yrow = a.Y + (size_t)row * k;
    if (singular) {
      if (tid == 0) atomicAdd(a.fail_counter + 1, 1);
      for (int e = tid; e < k; e += 256) yrow[e] = 0.f;
      continue;
    }
    // back substitution U y = b (one wave; sB becomes y)
    if (wv == 0) {
      for (int i = k - 1; i >= 0; i--) {
        float s = 0.f;
        for (int m = i + 1 + lane; m < k; m += 64) s = fmaf(sA[i * ld + m], sB[m], s);
        s = wave_sum(s);
        if (lane == 0) sB[i] = (sB[i] - s) / sA[i * ld + i];
        wave_sync();
      }
    }
    __syncthreads();
    for (int e = tid; e < k; e += 256) yrow[e] = sB[e];
    // loss row term (wrmf_implicit.hpp:257-270 / wrmf_explicit.hpp:131-132)
    if (wv == 0) {
      float lacc = 0.f;
      for (int j = 0; j < cnt; j++) {
        const float* xv = a.X + (size_t)a.row_idx[p1 + j] * k;
        float t = 0.f;
        for (int e = lane; e < k; e += 64) t = fmaf(xv[e], sB[e], t);
        t = wave_sum(t);
        const float c = a.vals[p1 + j];
        const float d = IMPLICIT ? (a.loss_tgt ? a.loss_tgt[p1 + j] : a.loss_tgt_const) - t : c - t;
        lacc += IMPLICIT ? c * d * d : d * d;
      }
      float yy = 0.f;
      for (int e = lane; e < k; e += 64) yy = fmaf(sB[e], sB[e], yy);
      yy = wave_sum(yy);
      if (lane == 0) wloss += IMPLICIT ? (double)lacc + a.lambda_loss * (double)yy : (double)(lacc + lam_use * yy);
    }
  }
  if (tid == 0) a.loss_partials[loss_slot0 + blockIdx.x] = wloss;
}

// start of a Cholesky half-iteration: the previous call's counts move to the running totals (words 2, 3), the list is empty again
__global__ void fail_roll_kernel(int* fails) {
  fails[2] += min(fails[0], kFailCap);                   // rows the general solver took
  fails[3] += fails[1] + max(0, fails[0] - kFailCap);   // rows beyond the list's capacity were never handed over: unresolved
  fails[0] = 0;
  fails[1] = 0;
}

}  // namespace

hipError_t launch_fail_roll(int* fails, hipStream_t s) {
  hipLaunchKernelGGL(fail_roll_kernel, dim3(1), dim3(1), 0, s, fails);
  return hipGetLastError();
}

// The rows the Cholesky launches of this half-iteration appended to a.fail_rows (count in a.fail_counter[0]); loss partials
// of its kLuGrid workgroups from loss_slot0 on (zero when there was nothing to do).
hipError_t launch_als_lu_fallback(const AlsArgs& a, bool implicit, size_t loss_slot0, hipStream_t s) {
  const size_t lds = ((size_t)a.k * (a.k + 1) + (size_t)a.k + 8 * (size_t)a.k + 16 + 2 + 8) * 4 + 64;
  auto ki = als_lu_fallback_kernel<true>;
  auto ke = als_lu_fallback_kernel<false>;
  hipError_t err = hipFuncSetAttribute(implicit ? reinterpret_cast<const void*>(ki) : reinterpret_cast<const void*>(ke),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return err;
  if (implicit) hipLaunchKernelGGL(ki, dim3(kLuGrid), dim3(256), lds, s, a, loss_slot0);
  else hipLaunchKernelGGL(ke, dim3(kLuGrid), dim3(256), lds, s, a, loss_slot0);
  return hipGetLastError();
}

}  // namespace rsparse_hip
