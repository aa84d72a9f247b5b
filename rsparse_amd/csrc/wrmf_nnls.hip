// Non-negative half-iteration: sequential coordinate descent NNLS per row (gfx950, wave64).
//
// Replaces the solver == SEQ_COORDINATE_WISE_NNLS branch of als_implicit<T> / als_explicit<T>
// (inst/include/wrmf_implicit.hpp:233-234, inst/include/wrmf_explicit.hpp:109-110) and c_nnls / scd_ls_update
// (inst/include/nnls.hpp:10-48):
//     lhs, rhs as in the Cholesky branch;  XtX = lhs^T lhs (+1e-16 on the diagonal);  mu = XtX init - lhs^T rhs;
//     sweeps over the coordinates in order:  new = max(0, h_k - mu_k / XtX_kk);  mu += (new - h_k) XtX[:,k]
//     until the largest relative coordinate change of a sweep is <= 1e-4 (SCD_TOL) or 10000 sweeps (SCD_MAX_ITER).
// The coordinate loop is a serial recurrence (Gauss-Seidel order is part of the result), so the parallelism is
// across rows (one 256-thread workgroup per row) and inside a step (the 128-long axpy on one wave):
//   assembly   as the Cholesky kernel: thread (I, K) of a 16 x 16 grid accumulates its BS x BS block in registers
//              from 32-vector chunks staged in LDS, then lhs is written to LDS in full (both triangles)
//   square     XtX block (I, K) = sum_m lhs[m][I-block] (x) lhs[m][K-block]  (lhs is symmetric: row reads only)
//   descend    wave 0: mu, h, diag live in registers (lane l owns coordinates l and l + 64); per coordinate two
//              v_readlane, a handful of scalar-like VALU ops, and one axpy with row k of XtX from LDS
//   loss       second pass over the row's chunks, as the Cholesky kernel
// LDS: two KP x KP matrices (128 KB at KP = 128) + the gather tile -> one workgroup per CU at rank 128.
#include "wrmf_internal.h"
#include "wrmf_device.h"

namespace rsparse_hip {
namespace {

using namespace dev;

constexpr int kScdMaxIter = 10000;   // SCD_MAX_ITER, inst/include/wrmf.hpp:20
constexpr float kScdTol = 1e-4f;     // SCD_TOL, inst/include/wrmf.hpp:21
constexpr float kNnlsEps = 1e-16f;   // EPS, inst/include/nnls.hpp:8

template <int KP>
struct NnlsSmem {
  static constexpr int BS = KP / 16;
  static constexpr int TC = 32;
  static constexpr int LDT = KP + 4;
  static constexpr size_t tile_floats = (size_t)TC * LDT;
  static constexpr size_t mat_floats = (size_t)KP * KP;
  static constexpr size_t vec_floats = (size_t)3 * KP + 2 * TC;  // rhs, init/result, spare, c, c1
  static constexpr size_t bytes = (tile_floats + 2 * mat_floats + vec_floats + 16) * 4 + 64;
};

template <int KP, bool VEC>
__device__ __forceinline__ void nnls_gather_chunk(const AlsArgs& a, int base, int ccnt, float* sT, int wv, int lane) {
  constexpr int LDT = KP + 4;
  const int k = a.k;
  if constexpr (VEC) {
    constexpr int LPV = KP / 4, VPI = 64 / LPV, NQ = 8 / VPI;
    const int c4 = lane % LPV, jo = lane / LPV;
    int ids[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) ids[q] = a.row_idx[base + min(8 * wv + q * VPI + jo, ccnt - 1)];
    float4 v[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) v[q] = *reinterpret_cast<const float4*>(a.X + (size_t)ids[q] * k + min(c4 * 4, k - 4));
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int j = 8 * wv + q * VPI + jo;
      if (j < ccnt && c4 * 4 < k) *reinterpret_cast<float4*>(sT + j * LDT + c4 * 4) = v[q];
    }
  } else {
    for (int j = 8 * wv; j < min(8 * wv + 8, ccnt); j++) {
      const int id = rfl(a.row_idx[base + j]);
      const float* src = a.X + (size_t)id * k;
      for (int e = lane; e < k; e += 64) sT[j * LDT + e] = src[e];
    }
  }
}

template <int KP, bool IMPLICIT, bool VEC>
__global__ __launch_bounds__(256) void als_nnls_kernel(AlsArgs a) {
  using SM = NnlsSmem<KP>;
  constexpr int BS = SM::BS, TC = SM::TC, LDT = SM::LDT, NS = KP / 64;  // NS coordinates per lane of wave 0
  static_assert(KP % 64 == 0 || KP == 32, "rank padding");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sT = reinterpret_cast<float*>(smem);
  float* sA = sT + SM::tile_floats;   // lhs, full symmetric, row-major
  float* sB = sA + SM::mat_floats;    // XtX = lhs^2
  float* sR = sB + SM::mat_floats;    // [KP] rhs
  float* sH = sR + KP;                // [KP] warm start -> result
  float* sC = sH + 2 * KP;            // [TC] confidence / rating
  float* sC1 = sC + TC;               // [TC] rank-one weight
  double* sLoss = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(sC1 + TC + 2) + 7) & ~(uintptr_t)7);

  const int tid = threadIdx.x, lane = tid & 63, wv = rfl(tid >> 6);
  const int I = tid >> 4, K = tid & 15;
  const bool lower = I >= K;
  const int k = a.k;
  for (int e = tid; e < (int)SM::tile_floats; e += 256) sT[e] = 0.f;
  __syncthreads();
  double wloss = 0.0;

  for (int row = blockIdx.x; row < a.n_cols; row += gridDim.x) {
    const int p1 = rfl(a.col_ptrs[row]), p2 = rfl(a.col_ptrs[row + 1]);
    const int cnt = p2 - p1;
    float* yrow = a.Y + (size_t)row * k;
    if (cnt <= 0 && !a.rhs_init) {  // empty column -> zeros (wrmf_implicit.hpp:281, wrmf_explicit.hpp:142)
      for (int e = tid; e < k; e += 256) yrow[e] = 0.f;
      continue;
    }
    const float lam_use =
        IMPLICIT ? 0.f : (float)(a.lambda_loss * (a.dynamic_lambda ? (double)(float)cnt : 1.0));

    // ---------------- assembly: lhs block (I, K) in registers, rhs[tid] ----------------
    float acc[BS][BS];
#pragma unroll
    for (int x = 0; x < BS; x++)
#pragma unroll
      for (int y = 0; y < BS; y++) {
        const int rr = I * BS + x, cc = K * BS + y;
        float gv;
        if (rr >= k || cc >= k) gv = (rr == cc) ? 1.f : 0.f;   // padded coordinates: identity, rhs 0 -> stay 0
        else if (IMPLICIT) gv = lower ? a.XtX[(size_t)rr * k + cc] : 0.f;
        else gv = (rr == cc) ? lam_use : 0.f;
        acc[x][y] = gv;
      }
    float rhs = 0.f;
    for (int base = p1; base < p2; base += TC) {
      const int ccnt = min(TC, p2 - base);
      __syncthreads();
      if (tid < ccnt) {
        const float cvv = a.vals[base + tid];
        sC[tid] = a.rhs_vals ? a.rhs_vals[base + tid] : cvv;   // coefficient in the right-hand side
        sC1[tid] = IMPLICIT ? cvv - 1.f : 1.f;
      }
      nnls_gather_chunk<KP, VEC>(a, base, ccnt, sT, wv, lane);
      __syncthreads();
      if (lower) {
        for (int j = 0; j < ccnt; j++) {
          const float c1 = sC1[j];
          float av[BS], bv[BS];
#pragma unroll
          for (int x = 0; x < BS; x++) {
            av[x] = sT[j * LDT + I * BS + x];
            bv[x] = sT[j * LDT + K * BS + x] * c1;
          }
#pragma unroll
          for (int x = 0; x < BS; x++)
#pragma unroll
            for (int y = 0; y < BS; y++) acc[x][y] = fmaf(av[x], bv[y], acc[x][y]);
        }
      }
      if (tid < KP) {
        float s = 0.f;
        for (int j = 0; j < ccnt; j++) s = fmaf(sC[j], sT[j * LDT + tid], s);
        rhs += s;
      }
    }
    __syncthreads();
    if (lower) {  // publish lhs, both triangles
#pragma unroll
      for (int x = 0; x < BS; x++)
#pragma unroll
        for (int y = 0; y < BS; y++) {
          sA[(I * BS + x) * KP + K * BS + y] = acc[x][y];
          if (I != K) sA[(K * BS + y) * KP + I * BS + x] = acc[x][y];
        }
    }
    if (tid < KP) {
      sR[tid] = rhs + ((a.rhs_init && tid < k) ? a.rhs_init[tid] : 0.f);
      sH[tid] = tid < k ? yrow[tid] : 0.f;  // init = current Y.col(i)  (wrmf_implicit.hpp:185)
    }
    __syncthreads();
    if (I == K) {  // the diagonal blocks hold their strict upper part as zeros from the init: mirror inside the block
#pragma unroll
      for (int x = 0; x < BS; x++)
#pragma unroll
        for (int y = 0; y < BS; y++)
          if (y > x) sA[(I * BS + x) * KP + K * BS + y] = acc[y][x];
    }
    __syncthreads();

    // ---------------- XtX = lhs^T lhs  (nnls.hpp:41-44), block (I, K) per thread ----------------
#pragma unroll
    for (int x = 0; x < BS; x++)
#pragma unroll
      for (int y = 0; y < BS; y++) acc[x][y] = 0.f;
    for (int m = 0; m < KP; m++) {
      float av[BS], bv[BS];
#pragma unroll
      for (int x = 0; x < BS; x++) {
        av[x] = sA[m * KP + I * BS + x];
        bv[x] = sA[m * KP + K * BS + x];
      }
#pragma unroll
      for (int x = 0; x < BS; x++)
#pragma unroll
        for (int y = 0; y < BS; y++) acc[x][y] = fmaf(av[x], bv[y], acc[x][y]);
    }
#pragma unroll
    for (int x = 0; x < BS; x++)
#pragma unroll
      for (int y = 0; y < BS; y++)
        sB[(I * BS + x) * KP + K * BS + y] = acc[x][y] + ((I == K && x == y) ? kNnlsEps : 0.f);
    __syncthreads();

    // ---------------- mu = XtX init - lhs^T rhs ; coordinate descent on wave 0 ----------------
    if (wv == 0) {
      float mu[NS > 0 ? NS : 1], h[NS > 0 ? NS : 1], dg[NS > 0 ? NS : 1];
      constexpr int NSL = KP >= 64 ? KP / 64 : 1;
#pragma unroll
      for (int s2 = 0; s2 < NSL; s2++) {
        const int c = lane + 64 * s2;
        float m = 0.f;
        if (c < KP) {
          for (int r = 0; r < KP; r++) m = fmaf(sB[r * KP + c], sH[r], fmaf(-sA[r * KP + c], sR[r], m));
        }
        mu[s2] = m;
        h[s2] = c < KP ? sH[c] : 0.f;
        dg[s2] = c < KP ? sB[c * KP + c] : 1.f;
      }
      // A coordinate that sits at its bound (h = 0) with a non-negative gradient does not move: new = max(0, 0 - mu / d) = 0,
      // diff = 0 (d > 0), and the reference's loop body does nothing for it (nnls.hpp:24).  Three quarters of the coordinate
      // visits are of that kind (measured on the config-2 shape: 172 sweeps per row, 23 % of the visits move), so the sweep
      // walks only the set bits of `act` = the lanes whose coordinate can move, re-evaluated after every move (a move changes
      // every mu): exactly the reference's sequence of updates, without the division and the three lane reads per idle visit.
      for (int t = 0; t < kScdMaxIter; t++) {
        float rel = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < NSL; s2++) {
          const int lim = min(64, min(KP, k) - 64 * s2);
          if (lim <= 0) break;  // padded coordinates never move
          const unsigned long long in_range = lim >= 64 ? ~0ull : ((1ull << lim) - 1ull);
          unsigned long long act = __ballot(!(h[s2] == 0.f && mu[s2] >= 0.f)) & in_range;
          while (act) {
            const int l = __builtin_ctzll(act);
            const int kk = 64 * s2 + l;
            const float old_v = readlane_f(h[s2], l);
            const float m_k = readlane_f(mu[s2], l);
            const float d_k = readlane_f(dg[s2], l);
            float new_v = old_v - m_k / d_k;
            if (new_v < 0.f) new_v = 0.f;
            const float diff = new_v - old_v;
            const unsigned long long above = l >= 63 ? 0ull : (~0ull << (l + 1));
            if (diff != 0.f) {  // wave-uniform
              if (lane == l) h[s2] = new_v;
              const float* brow = sB + kk * KP;  // column kk = row kk (symmetric)
#pragma unroll
              for (int s3 = 0; s3 < NSL; s3++) {
                const int c = lane + 64 * s3;
                if (c < KP) mu[s3] = fmaf(diff, brow[c], mu[s3]);
              }
              const float step_err = fabsf(diff) / (fabsf(old_v) + kNnlsEps);
              rel = fmaxf(rel, step_err);
              act = __ballot(!(h[s2] == 0.f && mu[s2] >= 0.f)) & in_range & above;
            } else {
              act &= above;
            }
          }
        }
        if (rel <= kScdTol) break;
      }
#pragma unroll
      for (int s2 = 0; s2 < NSL; s2++) {
        const int c = lane + 64 * s2;
        if (c < KP) sH[c] = h[s2];
      }
    }
    __syncthreads();
    if (tid < k) yrow[tid] = sH[tid];

    // ---------------- loss row term ----------------
    {
      float lacc = 0.f;
      for (int base = p1; base < p2; base += TC) {
        const int ccnt = min(TC, p2 - base);
        __syncthreads();
        if (tid < ccnt) {
          sC[tid] = a.vals[base + tid];
          sC1[tid] = a.loss_tgt ? a.loss_tgt[base + tid] : a.loss_tgt_const;
        }
        nnls_gather_chunk<KP, VEC>(a, base, ccnt, sT, wv, lane);
        __syncthreads();
        if (wv == 0) {
          const float t = tile_dot<KP, TC>(sT, sH, lane);
          const int jl = lane % TC;
          const float cvv = sC[jl < ccnt ? jl : 0];
          const float d = IMPLICIT ? sC1[jl < ccnt ? jl : 0] - t : cvv - t;
          lacc += (jl < ccnt && lane < TC) ? (IMPLICIT ? cvv * d * d : d * d) : 0.f;
        }
      }
      if (wv == 0) {
        const float lpart = wave_sum(lacc);
        float xxp = 0.f;
        for (int e = lane; e < k; e += 64) xxp = fmaf(sH[e], sH[e], xxp);
        xxp = wave_sum(xxp);
        if (lane == 0)
          wloss += IMPLICIT ? (double)lpart + a.lambda_loss * (double)xxp : (double)(lpart + lam_use * xxp);
      }
    }
  }
  __syncthreads();
  if (lane == 0) sLoss[wv] = wloss;
  __syncthreads();
  if (tid == 0) a.loss_partials[blockIdx.x] = (sLoss[0] + sLoss[1]) + (sLoss[2] + sLoss[3]);
}

template <int KP, bool IMPLICIT, bool VEC>
hipError_t launch_nnls_t(const AlsArgs& a, hipStream_t s, hipEvent_t* ev) {
  using SM = NnlsSmem<KP>;
  hipError_t err;
  const int grid = (int)chol_loss_slots(a.n_cols);
  auto kc = als_nnls_kernel<KP, IMPLICIT, VEC>;
  if ((err = hipFuncSetAttribute(reinterpret_cast<const void*>(kc), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)SM::bytes)) != hipSuccess)
    return err;
  if (ev && (err = hipEventRecord(ev[0], s)) != hipSuccess) return err;
  prof_note(ev, reinterpret_cast<const void*>(kc));
  hipLaunchKernelGGL(kc, dim3(grid), dim3(256), SM::bytes, s, a);
  if ((err = hipGetLastError()) != hipSuccess) return err;
  if (ev) {
    if ((err = hipEventRecord(ev[1], s)) != hipSuccess) return err;
    if ((err = hipEventRecord(ev[2], s)) != hipSuccess) return err;
  }
  return hipSuccess;
}

}  // namespace

hipError_t launch_als_nnls(const AlsArgs& a, bool implicit, hipStream_t s, hipEvent_t* ev) {
  const int KP = padded_rank(a.k);
  const bool vec = (a.k % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.X) & 15) == 0);
#define RSP_DISPATCH(KPV)                                                                                \
  if (KP == KPV) {                                                                                       \
    if (implicit) return vec ? launch_nnls_t<KPV, true, true>(a, s, ev) : launch_nnls_t<KPV, true, false>(a, s, ev); \
    return vec ? launch_nnls_t<KPV, false, true>(a, s, ev) : launch_nnls_t<KPV, false, false>(a, s, ev); \
  }
  RSP_DISPATCH(32)
  RSP_DISPATCH(64)
  RSP_DISPATCH(128)
#undef RSP_DISPATCH
  return hipErrorInvalidValue;
}

}  // namespace rsparse_hip
