// Ranks 129..256, conjugate gradient without bias operands: one WAVE per row, the operator applied from the gathered vectors as the
// reference applies it (gfx950, wave64; round 6).
//
// cg_solver_implicit / cg_solver_explicit (inst/include/wrmf_implicit.hpp:8-32, inst/include/wrmf_explicit.hpp:8-31):
//     A v = XtX v + X_nnz ((c - 1) o (X_nnz^T v))          (explicit: X_nnz X_nnz^T v + lambda_use v)
//     r0 = X_nnz (c - (c - 1) o X_nnz^T x0) - XtX x0        (explicit: X_nnz (r - X_nnz^T x0) - lambda_use x0)
// wrmf_wide.hip assembles every row's k x k system in LDS (k^2 flops per non-zero, one workgroup per row, one workgroup per CU at
// rank 256): its conjugate-gradient half-iteration cost 30 x what rank 128 costs -- and rank 128 WITH user/item biases is a system of
// order 129, i.e. this family (422 ms per iteration at 1M x 100k against 13.4 without the biases, profiles/r06/r6sweep_*).  Conjugate
// gradient never needed the matrix: 8 k (cg_steps + 1) flops per non-zero.  This is the fp32 sibling of f64_cg_wave_kernel
// (wrmf_f64.hip): lane l holds the coordinates l, 64 + l, 128 + l, 192 + l of every vector (EPL = 3 up to rank 192, else 4);
// a pass over the row gathers 16 non-zeros per batch (16 EPL loads per lane in flight), takes their 16 dot products with ONE
// transposed reduction (15 exchanges: lane L ends with the sum of non-zero L >> 2), forms the 16 coefficients at once and hands
// them back lane by lane for the sum over the vectors.  cg_steps + 2 passes per row (first residual, the steps, the loss) over
// vectors that are L2 / MALL resident from the second pass on.  XtX (implicit): the packed lower triangle in LDS, shared by the
// workgroup's four waves.  Double scalars rsold / alpha / beta (:18), the 1e-10 exit (:27).
// Rows beyond n_hi non-zeros stay with wrmf_wide.hip's kernel (a workgroup per row: one wave would walk a 1e5-non-zero row alone).
#include <algorithm>

#include "wrmf_internal.h"
#include "wrmf_device.h"

namespace rsparse_hip {
namespace {

using namespace dev;

constexpr float kCgTolWc = 1e-10f;   // CG_TOL, inst/include/wrmf.hpp:22

__device__ __forceinline__ int tri_wc(const int i) { return (i * (i + 1)) >> 1; }

// sixteen per-lane partial sums -> lane L holds the wave's sum of part[(L >> 2) & 15] (the four lanes of a quad hold copies):
// every stage halves the number of values a lane carries by exchanging the half it does not keep with the lane across
__device__ __forceinline__ float transposed_sum16(float (&v)[16], const int lane) {
#pragma unroll
  for (int st = 0; st < 4; st++) {
    const int m = 32 >> st, n = 8 >> st;   // lane bit 5, 4, 3, 2 <-> value bit 3, 2, 1, 0
    const bool up = (lane & m) != 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (i < n) {
        const float keep = up ? v[i + n] : v[i];
        const float send = up ? v[i] : v[i + n];
        v[i] = keep + __shfl_xor(send, m);
      }
    }
  }
  float s = v[0];
  s += __shfl_xor(s, 2);
  s += __shfl_xor(s, 1);
  return s;
}

template <int EPL>
struct WVec {
  float c[EPL];
};

struct WideCgArgs {
  const int32_t* col_ptrs;
  const int32_t* row_idx;
  const float* vals;
  const float* X;
  float* Y;
  const float* XtX;
  int n_cols, k, cg_steps, dynamic_lambda, n_hi;
  double lambda_loss;
  double* loss_partials;
  const int32_t* order;   // TEAM > 1: the rows longest first (DevCSC::q_order); the launch takes those beyond n_hi non-zeros
};

// MODE 0: first residual's gather term, 1: operator's, 2: the loss sum (returned in c[0] of every lane)
// TEAM > 1: the workgroup's TEAM waves share the row -- wave w takes the chunks w, w + TEAM, ... --, their partial results meet in
// LDS (sT: TEAM x 256 floats) and every wave leaves with the same bits (summed in wave order)
template <int EPL, bool IMPLICIT, int MODE, int TEAM>
__device__ __forceinline__ WVec<EPL> wide_row_pass(const WideCgArgs& a, const int k, const int p1, const int n, const WVec<EPL>& v,
                                                   const bool (&lk)[EPL], const int (&lc)[EPL], const int lane, const int wv, float* sT) {
  WVec<EPL> acc;
#pragma unroll
  for (int e = 0; e < EPL; e++) acc.c[e] = 0.f;
  for (int c0 = 64 * (TEAM > 1 ? wv : 0); c0 < n; c0 += 64 * TEAM) {
    const int cn = min(64, n - c0);
    const int mine = min(lane, cn - 1);
    const int idj = a.row_idx[p1 + c0 + mine];
    const float cj = a.vals[p1 + c0 + mine];
    for (int b0 = 0; b0 < cn; b0 += 16) {   // wave-uniform
      float cur[16][EPL];
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const int id = __builtin_amdgcn_readlane(idj, min(b0 + u, cn - 1));   // (beyond the chunk: its last non-zero again, weight 0)
        const float* xr = a.X + (size_t)id * k;
#pragma unroll
        for (int e = 0; e < EPL; e++) cur[u][e] = xr[lc[e]];
      }
      float part[16];
#pragma unroll
      for (int u = 0; u < 16; u++) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; e++) {
          cur[u][e] = lk[e] ? cur[u][e] : 0.f;
          s = fmaf(cur[u][e], v.c[e], s);
        }
        part[u] = s;
      }
      const float t = transposed_sum16(part, lane);
      const int st = b0 + ((lane >> 2) & 15);   // the lane's non-zero of the chunk
      const float c = __shfl(cj, min(st, cn - 1));
      const bool in = st < cn;
      if constexpr (MODE == 2) {
        const float dlt = (IMPLICIT ? 1.f : c) - t;
        acc.c[0] += (in && (lane & 3) == 0) ? (IMPLICIT ? c : 1.f) * dlt * dlt : 0.f;   // (one lane of the quad counts it)
      } else {
        float coef;
        if constexpr (MODE == 0) coef = IMPLICIT ? c - (c - 1.f) * t : c - t;
        else coef = IMPLICIT ? (c - 1.f) * t : t;
        coef = in ? coef : 0.f;
#pragma unroll
        for (int u = 0; u < 16; u++) {
          const float cu = readlane_f(coef, 4 * u);
#pragma unroll
          for (int e = 0; e < EPL; e++) acc.c[e] = fmaf(cu, cur[u][e], acc.c[e]);
        }
      }
    }
  }
  if constexpr (MODE == 2) acc.c[0] = wave_sum(acc.c[0]);
  if constexpr (TEAM > 1) {
    __syncthreads();   // (the previous pass's readers are done with sT)
    if constexpr (MODE == 2) {
      if (lane == 0) sT[256 * wv] = acc.c[0];
    } else {
#pragma unroll
      for (int e = 0; e < EPL; e++) sT[256 * wv + lane + 64 * e] = acc.c[e];
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < (MODE == 2 ? 1 : EPL); e++) {
      float sum = 0.f;
      for (int w = 0; w < TEAM; w++) sum += sT[256 * w + (MODE == 2 ? 0 : lane + 64 * e)];
      acc.c[e] = sum;
    }
  }
  return acc;
}

template <int EPL, bool IMPLICIT, int TEAM>
__global__ __launch_bounds__(TEAM > 1 ? 64 * TEAM : 256, TEAM > 1 ? 1 : 2) void wide_cg_wave_kernel(WideCgArgs a, int slot0) {
  constexpr int NTW = TEAM > 1 ? 64 * TEAM : 256;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sT = reinterpret_cast<float*>(smem_raw);                    // TEAM > 1: the waves' partial results
  float* sG = sT + (TEAM > 1 ? 256 * TEAM : 0);                      // XtX (implicit): packed lower triangle, row i at i (i + 1) / 2
  __shared__ double sLoss[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = rfl(tid >> 6);
  const int k = a.k;
  bool lk[EPL];
  int lc[EPL], tl[EPL];
#pragma unroll
  for (int e = 0; e < EPL; e++) {
    lk[e] = lane + 64 * e < k;
    lc[e] = min(lane + 64 * e, k - 1);
    tl[e] = tri_wc(lc[e]);
  }
  if (IMPLICIT) {
    for (int e = tid; e < k * k; e += NTW) {
      const int i = e / k, j = e - i * k;
      if (j <= i) sG[tri_wc(i) + j] = a.XtX[e];
    }
    __syncthreads();
  }
  using Vec = WVec<EPL>;
  // (G v)_l for the lane's coordinates; v_m comes from lane m % 64, register m / 64
  auto gmv = [&](const Vec& v) {
    Vec s0;
#pragma unroll
    for (int e = 0; e < EPL; e++) s0.c[e] = 0.f;
#pragma unroll
    for (int e2 = 0; e2 < EPL; e2++) {
      const int lim = min(64, k - 64 * e2);
      for (int m0 = 0; m0 < lim; m0 += 8) {   // eight columns per trip, their LDS reads in flight together
        float gv[8][EPL];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int mm = 64 * e2 + min(m0 + u, lim - 1);
          const int tm = tri_wc(mm);
#pragma unroll
          for (int e = 0; e < EPL; e++) gv[u][e] = sG[lc[e] >= mm ? tl[e] + mm : tm + lc[e]];   // symmetric: (row, col) with row >= col
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const float vm = m0 + u < lim ? readlane_f(v.c[e2], m0 + u) : 0.f;
#pragma unroll
          for (int e = 0; e < EPL; e++) s0.c[e] = fmaf(gv[u][e], vm, s0.c[e]);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < EPL; e++) s0.c[e] = lk[e] ? s0.c[e] : 0.f;
    return s0;
  };
  auto dot = [&](const Vec& u, const Vec& w) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; e++) s = fmaf(lk[e] ? u.c[e] : 0.f, w.c[e], s);
    return wave_sum(s);
  };
  double wloss = 0.0;
  for (int it = TEAM > 1 ? blockIdx.x : blockIdx.x * 4 + wv; it < a.n_cols; it += TEAM > 1 ? gridDim.x : gridDim.x * 4) {
    const int row = TEAM > 1 ? a.order[it] : it;
    const int p1 = a.col_ptrs[row], n = a.col_ptrs[row + 1] - p1;
    float* yrow = a.Y + (size_t)row * k;
    if constexpr (TEAM > 1) {
      if (n <= a.n_hi) break;     // (longest first: nothing beyond is this launch's; workgroup-uniform)
    } else {
      if (n > a.n_hi) continue;   // the team launch takes it
    }
    if (n <= 0) {   // empty column -> zeros (wrmf_implicit.hpp:272-283, wrmf_explicit.hpp:133-144)
#pragma unroll
      for (int e = 0; e < EPL; e++)
        if (lane + 64 * e < k) yrow[lane + 64 * e] = 0.f;
      continue;
    }
    const float lam_use = IMPLICIT ? (float)a.lambda_loss : (float)(a.lambda_loss * (a.dynamic_lambda ? (double)n : 1.0));
    Vec x, r, pv;
#pragma unroll
    for (int e = 0; e < EPL; e++) x.c[e] = lk[e] ? yrow[lane + 64 * e] : 0.f;   // warm start
    {
      const Vec t0 = wide_row_pass<EPL, IMPLICIT, 0, TEAM>(a, k, p1, n, x, lk, lc, lane, wv, sT);
      Vec g0;
      if constexpr (IMPLICIT) g0 = gmv(x);
#pragma unroll
      for (int e = 0; e < EPL; e++) r.c[e] = t0.c[e] - (IMPLICIT ? g0.c[e] : lam_use * x.c[e]);
    }
    pv = r;
    double rsold = (double)dot(r, r);
    for (int it = 0; it < a.cg_steps; it++) {
      Vec ap = wide_row_pass<EPL, IMPLICIT, 1, TEAM>(a, k, p1, n, pv, lk, lc, lane, wv, sT);
      {
        Vec g1;
        if constexpr (IMPLICIT) g1 = gmv(pv);
#pragma unroll
        for (int e = 0; e < EPL; e++) ap.c[e] += IMPLICIT ? g1.c[e] : lam_use * pv.c[e];
      }
      const float alpha = (float)(rsold / (double)dot(pv, ap));
#pragma unroll
      for (int e = 0; e < EPL; e++) {
        x.c[e] = fmaf(alpha, pv.c[e], x.c[e]);
        r.c[e] = fmaf(-alpha, ap.c[e], r.c[e]);
      }
      const double rsnew = (double)dot(r, r);
      if (rsnew < (double)kCgTolWc) break;
      const float beta = (float)(rsnew / rsold);
#pragma unroll
      for (int e = 0; e < EPL; e++) pv.c[e] = fmaf(pv.c[e], beta, r.c[e]);
      rsold = rsnew;
    }
    if (TEAM == 1 || wv == 0) {
#pragma unroll
      for (int e = 0; e < EPL; e++)
        if (lane + 64 * e < k) yrow[lane + 64 * e] = x.c[e];
    }
    const Vec lrow = wide_row_pass<EPL, IMPLICIT, 2, TEAM>(a, k, p1, n, x, lk, lc, lane, wv, sT);
    const float xx = dot(x, x);
    wloss += IMPLICIT ? (double)lrow.c[0] + a.lambda_loss * (double)xx : (double)(lrow.c[0] + lam_use * xx);
  }
  if constexpr (TEAM > 1) {   // (every wave holds the same sum)
    if (tid == 0) a.loss_partials[slot0 + blockIdx.x] = wloss;
  } else {
    if (lane == 0) sLoss[wv] = wloss;
    __syncthreads();
    if (tid == 0) a.loss_partials[slot0 + blockIdx.x] = (sLoss[0] + sLoss[1]) + (sLoss[2] + sLoss[3]);
  }
}

}  // namespace

// plain conjugate gradient at the wide ranks (no per-non-zero operands, no global bias)
bool wide_cg_wave_supported(const AlsArgs& a, unsigned solver) {
  return solver == 1 && a.k > 128 && a.k <= 256 && !a.rhs_vals && !a.loss_tgt && !a.rhs_init && a.gbias == 0.f;
}
int wide_cg_wave_grid(int n_cols) { return std::max(1, std::min((n_cols + 3) / 4, 256 * 16)); }
int wide_cg_team_grid(int n_cols) { return std::max(1, std::min(n_cols, 1024)); }
constexpr int kWideCgTeam = 8;

// Two launches: the rows beyond n_hi non-zeros (order = the rows longest first; nullptr: there are none) on teams of 8 waves,
// a workgroup each -- first: they end the half-iteration --, then the rest one wave per row.  Loss partials
// a.loss_partials[slot0 .. slot0 + wide_cg_team_grid + wide_cg_wave_grid)
hipError_t launch_wide_cg_wave(const AlsArgs& a, bool implicit, int n_hi, const int32_t* order, int slot0, hipStream_t s) {
  if (a.n_cols <= 0) return hipSuccess;
  WideCgArgs w;
  w.col_ptrs = a.col_ptrs; w.row_idx = a.row_idx; w.vals = a.vals; w.X = a.X; w.Y = a.Y; w.XtX = a.XtX;
  w.n_cols = a.n_cols; w.k = a.k; w.cg_steps = a.cg_steps; w.dynamic_lambda = a.dynamic_lambda; w.n_hi = n_hi;
  w.lambda_loss = a.lambda_loss; w.loss_partials = a.loss_partials; w.order = order;
  const int grid_t = wide_cg_team_grid(a.n_cols), grid = wide_cg_wave_grid(a.n_cols);
  const size_t ldsG = implicit ? (size_t)a.k * (a.k + 1) / 2 * sizeof(float) : 0;
  hipError_t err;
  if (!order) {
    if ((err = hipMemsetAsync(a.loss_partials + slot0, 0, (size_t)grid_t * sizeof(double), s)) != hipSuccess) return err;
  }
#define RSP_WCG(EPL, IMPL, TEAM, GRID, SLOT)                                                                                     \
  {                                                                                                                              \
    auto kern = wide_cg_wave_kernel<EPL, IMPL, TEAM>;                                                                            \
    const size_t lds = ldsG + (TEAM > 1 ? (size_t)256 * TEAM * sizeof(float) : 0);                                               \
    if (lds > 48 * 1024 &&                                                                                                       \
        (err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) \
      return err;                                                                                                                \
    hipLaunchKernelGGL(kern, dim3(GRID), dim3(TEAM > 1 ? 64 * TEAM : 256), lds, s, w, SLOT);                                     \
  }
#define RSP_WCG2(EPL, IMPL)                                                      \
  {                                                                              \
    if (order) RSP_WCG(EPL, IMPL, kWideCgTeam, grid_t, slot0)                    \
    RSP_WCG(EPL, IMPL, 1, grid, slot0 + grid_t)                                  \
  }
  if (a.k <= 192) {
    if (implicit) RSP_WCG2(3, true) else RSP_WCG2(3, false)
  } else {
    if (implicit) RSP_WCG2(4, true) else RSP_WCG2(4, false)
  }
#undef RSP_WCG2
#undef RSP_WCG
  return hipGetLastError();
}

}  // namespace rsparse_hip
