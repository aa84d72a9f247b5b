// Exact (Cholesky-branch) solve for SHORT rows in low-rank form (gfx950, wave64).
//
// Same reference branch as wrmf_chol.hip (inst/include/wrmf_implicit.hpp:206-208,231,236: y = solve(lhs, rhs) with
// lhs = XtX + X_nnz diag(c - 1) X_nnz^T, rhs = X_nnz c), for rows with n <= 64 non-zeros and implicit feedback -- on the
// user side of the bench matrix that is almost every row, and there the k x k factorisation (k^3 / 3 flops, a serial
// chain of k pivots) is 57 % of wrmf_chol.hip's time although the row contributes only a rank-n update to XtX.
//
// With XtX = L L^T factored ONCE per half-iteration (chol_lr_prep_kernel) and M = L^-T:
//     lhs = L (I_k + W^T W) L^T,   W = D^1/2 V',  V' = X_nnz M  (n x k),  D = diag(c - 1)
//     (I_k + W^T W)^-1 = I_k - W^T S^-1 W,        S = I_n + W W^T  (n x n, eigenvalues >= 1)
//     y = M (g - W^T S^-1 W g),                   g = M^T rhs = V'^T c
// so the per-row work is two small GEMMs on the matrix cores (V' = X_nnz M and V' V'^T, exact fp32 MFMA), an n x n
// LDL^T instead of a k x k Cholesky, and a handful of matrix-vector products.  The loss needs no second gather:
// x_j . y = v_j . q with q = g - W^T S^-1 W g.  Any exact method satisfies the reference's `solve`; the parity bound
// (1e-4 against the fp64 oracle) is the same as for wrmf_chol.hip and is checked by the same tests.
//
// Needs every confidence >= 1 (D^1/2) and XtX positive definite: both are decided on the device (flags[0] != 0 ->
// this kernel returns at once and wrmf_chol.hip's kernel, which otherwise skips the short rows, takes them).
#include <type_traits>

#include "wrmf_internal.h"
#include "wrmf_device.h"

namespace rsparse_hip {
namespace {

using namespace dev;

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef RSP_LR_ABL
#define RSP_LR_ABL 0   // dev builds: timing-only ablations (1 no V' GEMM, 2 no S GEMM, 4 no LDL^T, 8 no substitution, 16 no y = M q, 32 no gather)
#endif
constexpr int kLrLd = 130;   // LDS row stride of the n x k operands: conflict-free for the MFMA operand reads (2 i + c)
constexpr int kLrLs = 65;    // ... of the n x n system

// ---- XtX = L L^T, M = L^-T, Mt = M^T = L^-1: one workgroup, once per half-iteration ----------------------------------
// M and Mt are written KP x KP, zero padded.  flags[0] |= 1 when XtX is not positive definite.
template <int KP>
__global__ __launch_bounds__(256) void chol_lr_prep_kernel(const float* __restrict__ G, int k, float* __restrict__ M,
                                                           float* __restrict__ Mt, unsigned* __restrict__ flags) {
  constexpr int LD = KP + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sL = reinterpret_cast<float*>(smem);   // [KP][LD]
  float* sZ = sL + KP * LD;                     // [KP][LD]  L^-1
  const int tid = threadIdx.x;
  for (int e = tid; e < KP * KP; e += 256) {
    const int r = e / KP, c = e % KP;
    sL[r * LD + c] = (r < k && c < k) ? G[(size_t)r * k + c] : (r == c ? 1.f : 0.f);
    sZ[r * LD + c] = 0.f;
  }
  // right-looking elimination without scaling the pivot column (one barrier per step): after step j the entries
  // (i, c), i >= c > j, hold the Schur complement; column j keeps its values of step j
  const int ti = tid >> 4, tc = tid & 15;
  bool bad = false;
  for (int j = 0; j < k; j++) {
    __syncthreads();
    const float d = sL[j * LD + j];
    if (!(d > 0.f)) bad = true;
    const float inv = 1.f / d;
    for (int i = j + 1 + ti; i < k; i += 16) {
      const float lij = sL[i * LD + j] * inv;
      for (int c = j + 1 + tc; c <= i; c += 16) sL[i * LD + c] -= lij * sL[c * LD + j];
    }
  }
  __syncthreads();
  if (bad && tid == 0) atomicOr(flags, 1u);
  // L[i][j] = S[i][j] / sqrt(S[j][j]); kept in place (column scaling), diagonal = sqrt
  for (int e = tid; e < k * k; e += 256) {
    const int i = e / k, j = e % k;
    if (i > j) sL[i * LD + j] = sL[i * LD + j] * rsqrtf(fmaxf(sL[j * LD + j], 1e-30f));
  }
  __syncthreads();
  for (int j = tid; j < k; j += 256) sL[j * LD + j] = sqrtf(fmaxf(sL[j * LD + j], 1e-30f));
  __syncthreads();
  // Z = L^-1, one column per thread (forward substitution on e_c), double accumulation
  if (tid < k) {
    const int c = tid;
    sZ[c * LD + c] = 1.f / sL[c * LD + c];
    for (int i = c + 1; i < k; i++) {
      double s = 0.0;
      for (int m = c; m < i; m++) s += (double)sL[i * LD + m] * (double)sZ[m * LD + c];
      sZ[i * LD + c] = (float)(-s / (double)sL[i * LD + i]);
    }
  }
  __syncthreads();
  for (int e = tid; e < KP * KP; e += 256) {
    const int r = e / KP, c = e % KP;
    const bool in = r < k && c < k;
    Mt[e] = (in && c <= r) ? sZ[r * LD + c] : 0.f;   // Mt = L^-1 (lower)
    M[e] = (in && r <= c) ? sZ[c * LD + r] : 0.f;    // M = L^-T (upper)
  }
}

// ---- the rows ----------------------------------------------------------------------------------------------------
template <int KP>
struct LrSmem {
  static constexpr int NP = 64;
  static constexpr size_t x_floats = (size_t)NP * kLrLd;   // X_nnz, later S (NP x kLrLs fits)
  static constexpr size_t v_floats = (size_t)NP * kLrLd;   // V'
  static constexpr size_t vec_floats = 4 * NP + 4 * KP + 64;
  static constexpr size_t bytes = (x_floats + v_floats + vec_floats) * 4 + 64;
};

template <int KP>
__global__ __launch_bounds__(256, 2) void als_chol_lr_kernel(AlsArgs a, const int32_t* __restrict__ rows, int n_rows,
                                                             const float* __restrict__ M, const float* __restrict__ Mt,
                                                             const unsigned* __restrict__ flags, int loss_slot0) {
  using SM = LrSmem<KP>;
  constexpr int NP = SM::NP, LD = kLrLd, LS = kLrLs, NKS = KP / 2;
  static_assert(KP == 128, "written for rank 97..128");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sX = reinterpret_cast<float*>(smem);
  float* sS = sX;                        // alias: X_nnz is dead once V' exists
  float* sV = sX + SM::x_floats;
  float* sC = sV + SM::v_floats;         // [NP] confidences
  float* sQ = sC + NP;                   // [NP] sqrt(c - 1)
  float* sH = sQ + NP;                   // [NP] h = W g, then sqrt(c - 1) z
  float* sT = sH + NP;                   // [NP] spare
  float* sGv = sT + NP;                  // [KP] g
  float* sQv = sGv + KP;                 // [KP] q
  float* sY = sQv + KP;                  // [KP] y (two partial halves are added through sP)
  float* sP = sY + KP;                   // [KP] second half of y
  double* sRed = reinterpret_cast<double*>(sP + KP);   // [8]
  const int tid = threadIdx.x, lane = tid & 63, wv = rfl(tid >> 6);
  const int col = lane & 31, half = lane >> 5;
  const int k = a.k;
  if (flags[0] != 0) return;   // some confidence < 1 or XtX not positive definite: wrmf_chol.hip takes these rows

  // B operand of V' = X_nnz M for this wave's column block: M[kk][32 wv + col], kk = 2 t + half; M is upper
  // triangular: nothing below row 32 (wv + 1).  Re-read from L2 for every row: keeping the 64 registers for the whole launch
  // (or across the gather) spills.

  double wloss = 0.0;
  // Row metadata runs ahead of the solves so that the gather of a row is ONE memory round trip (it was a chain of four:
  // row id -> pointers -> index -> vector, and that per vector): the row id three rows ahead, its pointers two ahead,
  // its indices and confidences (lane j holds non-zero j) one ahead.
  const int G = gridDim.x;
  int it = blockIdx.x;
  int row_c = 0, p1_c = 0, n_c = 0, row_n = 0, p1_n = 0, n_n = 0, row_nn = 0;
  if (it < n_rows) {
    row_c = rows[it];
    p1_c = a.col_ptrs[row_c];
    n_c = a.col_ptrs[row_c + 1] - p1_c;
  }
  if (it + G < n_rows) {
    row_n = rows[it + G];
    p1_n = a.col_ptrs[row_n];
    n_n = a.col_ptrs[row_n + 1] - p1_n;
  }
  if (it + 2 * G < n_rows) row_nn = rows[it + 2 * G];
  int id_c = 0;
  float c_c = 1.f;
  if (it < n_rows && lane < n_c) {
    id_c = a.row_idx[p1_c + lane];
    c_c = a.vals[p1_c + lane];
  }
  for (; it < n_rows; it += G) {
    const int row = rfl(row_c);
    const int p1 = rfl(p1_c), n = rfl(n_c);   // 1 <= n <= 64 (launcher)
    const int nrt = n <= 32 ? 1 : 2;   // 32-row tiles
    __syncthreads();                   // the previous row's buffers are free
    // requests for the rows to come (consumed at the bottom of this iteration)
    int id_nx = 0, p1_nn = 0, n_nn = 0, row_n3 = 0;
    float c_nx = 1.f;
    {
      const int p1n = rfl(p1_n), nn = rfl(n_n);
      if (it + G < n_rows && lane < nn) {
        id_nx = a.row_idx[p1n + lane];
        c_nx = a.vals[p1n + lane];
      }
      if (it + 2 * G < n_rows) {
        const int rnn = rfl(row_nn);
        p1_nn = a.col_ptrs[rnn];
        n_nn = a.col_ptrs[rnn + 1] - p1_nn;
      }
      if (it + 3 * G < n_rows) row_n3 = rows[it + 3 * G];
    }
    // 1. gather: wave w takes vectors w, w + 4, ...; a lane copies 2 floats of each.  All of a wave's loads are issued
    // before the first LDS store (slots past the row repeat its last vector and are stored as zeros).
    {
      float2 v[16];
      const bool on = !(RSP_LR_ABL & 32);
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int id = __builtin_amdgcn_readlane(id_c, min(wv + 4 * u, n - 1));
        v[u] = (on && 2 * lane < k) ? *reinterpret_cast<const float2*>(a.X + (size_t)id * k + 2 * lane) : float2{0.f, 0.f};
      }
      if (nrt == 2) {
#pragma unroll
        for (int u = 8; u < 16; u++) {
          const int id = __builtin_amdgcn_readlane(id_c, min(wv + 4 * u, n - 1));
          v[u] = (on && 2 * lane < k) ? *reinterpret_cast<const float2*>(a.X + (size_t)id * k + 2 * lane) : float2{0.f, 0.f};
        }
      }
      if (wv == 0 && lane < 32 * nrt) {
        const float c = on ? c_c : 1.f;
        sC[lane] = lane < n ? c : 0.f;
        sQ[lane] = lane < n ? sqrtf(fmaxf(c - 1.f, 0.f)) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int j = wv + 4 * u;
        *reinterpret_cast<float2*>(sX + j * LD + 2 * lane) = j < n ? v[u] : float2{0.f, 0.f};
      }
      if (nrt == 2) {
#pragma unroll
        for (int u = 8; u < 16; u++) {
          const int j = wv + 4 * u;
          *reinterpret_cast<float2*>(sX + j * LD + 2 * lane) = j < n ? v[u] : float2{0.f, 0.f};
        }
      }
    }
    __syncthreads();
    // 2. V' = X_nnz M on the matrix cores: this wave's 32 columns, all row tiles.  (The pointer is made opaque so that
    // the loads stay inside the row loop -- hoisted, they pin 64 registers for the whole launch -- and the triangular
    // cut-off is taken per chunk of 16 k-steps: one uniform branch per chunk instead of one per MFMA.)
    if (!(RSP_LR_ABL & 1)) {
      const float* Mp = M + (size_t)half * KP + 32 * wv + col;
      asm volatile("" : "+v"(Mp));
      float breg[4][16];
#pragma unroll
      for (int ch = 0; ch < 4; ch++)
        if (ch <= wv) {
#pragma unroll
          for (int t = 0; t < 16; t++) breg[ch][t] = Mp[(size_t)(2 * (16 * ch + t)) * KP];
        }
      for (int rt = 0; rt < nrt; rt++) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; e++) acc[e] = 0.f;
        const float* xa = sX + (32 * rt + col) * LD + half;
#pragma unroll
        for (int ch = 0; ch < 4; ch++)
          if (ch <= wv) {
#pragma unroll
            for (int t = 0; t < 16; t++)
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[2 * (16 * ch + t)], breg[ch][t], acc, 0, 0, 0);
          }
#pragma unroll
        for (int e = 0; e < 16; e++)
          sV[(32 * rt + (e & 3) + 8 * (e >> 2) + 4 * half) * LD + 32 * wv + col] = acc[e];
      }
    }
    __syncthreads();
    // 3. g = V'^T c
    if (tid < KP) {
      float s = 0.f;
      for (int j = 0; j < n; j++) s = fmaf(sC[j], sV[j * LD + tid], s);
      sGv[tid] = s;
    }
    __syncthreads();
    // 4. h = D^1/2 V' g  (4 threads per row)
    {
      const int j = tid >> 2, part = tid & 3;
      float s = 0.f;
      if (j < 32 * nrt) {
        const float* vr = sV + j * LD + 32 * part;
#pragma unroll 8
        for (int e = 0; e < 32; e++) s = fmaf(vr[e], sGv[32 * part + e], s);
      }
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      if (part == 0 && j < NP) sH[j] = s * sQ[j];
    }
    // 5. S = I + D^1/2 V' V'^T D^1/2 (lower tiles) on the matrix cores -> sS (over X_nnz)
    {
      const int rt = wv == 0 ? 0 : 1, ct = wv == 2 ? 1 : 0;
      if (!(RSP_LR_ABL & 2) && wv < (nrt == 1 ? 1 : 3)) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; e++) acc[e] = 0.f;
        const float* va = sV + (32 * rt + col) * LD + half;
        const float* vb = sV + (32 * ct + col) * LD + half;
#pragma unroll
        for (int t = 0; t < NKS; t++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(va[2 * t], vb[2 * t], acc, 0, 0, 0);
        const float qc = sQ[32 * ct + col];
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int i = 32 * rt + (e & 3) + 8 * (e >> 2) + 4 * half, c2 = 32 * ct + col;
          sS[i * LS + c2] = (i == c2 ? 1.f : 0.f) + sQ[i] * qc * acc[e];
        }
      }
    }
    __syncthreads();
    // 6 + 7. z = S^-1 h by ONE wave, in registers: lane i holds row i of the (full, symmetric) matrix.  Right-looking
    // LDL^T: at step j the pivot row is broadcast entry by entry (v_readlane of lane j's registers: S[c][j] = S[j][c]) and
    // every lane updates its own row; the Schur complements stay symmetric, so at the end lane i holds, left of the
    // diagonal, column values frozen at their pivot steps (row i of L times D) and, right of it, its own pivot row
    // (column i of L times d_i) -- both triangular solves read nothing but the lane's own registers and broadcast scalars.
    // The forward substitution rides along with the elimination.  No barriers, no LDS traffic after the row is loaded.
    if (!(RSP_LR_ABL & 4) && wv == 0) {
      int i = lane;
      asm volatile("" : "+v"(i));   // laundered per row: hipcc otherwise hoists the 64 load addresses below out of the row loop and spills them
      auto solve = [&](auto np_tag) {
        constexpr int NS = decltype(np_tag)::value;
        float r[NS];
#pragma unroll
        for (int c = 0; c < NS; c++) r[c] = (i < n && c < n) ? sS[(i >= c ? i * LS + c : c * LS + i)] : (i == c ? 1.f : 0.f);
        float u = i < n ? sH[i] : 0.f;
        float dinv = 1.f;   // 1 / d_i, set when row i is the pivot row
#pragma unroll
        for (int j = 0; j < NS; j++) {
          const float pj = readlane_f(r[j], j);   // >= 1 (S = I + W W^T)
          const float r0 = __builtin_amdgcn_rcpf(pj);
          const float inv = fmaf(fmaf(-pj, r0, 1.f), r0, r0);   // v_rcp_f32 + one Newton step: the pivots' serial spine
          const float uj = readlane_f(u, j);
          if (i == j) dinv = inv;
          const float lij = i > j ? r[j] * inv : 0.f;   // L_ij; rows <= j are finished
          u = fmaf(-lij, uj, u);
#pragma unroll
          for (int c = j + 1; c < NS; c++) r[c] = fmaf(-lij, readlane_f(r[c], j), r[c]);
        }
        // backward: z_c = (u_c - sum_{c' > c} d_c L[c'][c] z_c') / d_c, largest index first
        float acc = 0.f, z = 0.f;
#pragma unroll
        for (int c = NS - 1; c >= 0; c--) {
          if (i == c) z = (u - acc) * dinv;
          const float zc = readlane_f(z, c);
          acc = fmaf(i < c ? r[c] : 0.f, zc, acc);
        }
        sH[i] = i < n ? z * sQ[i] : 0.f;   // D^1/2 z
      };
      if (!(RSP_LR_ABL & 8)) {
        if (n <= 16) solve(std::integral_constant<int, 16>{});
        else if (n <= 32) solve(std::integral_constant<int, 32>{});
        else if (n <= 48) solve(std::integral_constant<int, 48>{});
        else solve(std::integral_constant<int, NP>{});
      }
    }
    __syncthreads();
    // 8. q = g - V'^T (D^1/2 z)
    if (tid < KP) {
      float s = sGv[tid];
      for (int j = 0; j < n; j++) s = fmaf(-sH[j], sV[j * LD + tid], s);
      sQv[tid] = s;
    }
    __syncthreads();
    // 9. y = M q = sum_j Mt[j][.] q_j  (rows of Mt are coalesced; the two halves of the workgroup split j)
    {
      const int i = tid & (KP - 1), hj = tid >> 7;
      const float* Mtp = Mt + (size_t)(64 * hj) * KP + i;
      asm volatile("" : "+v"(Mtp));   // (opaque: these loads must not be hoisted out of the row loop either)
      float s = 0.f;
#pragma unroll
      for (int j0 = 0; j0 < ((RSP_LR_ABL & 16) ? 0 : 64); j0 += 32) {   // 32 independent L2 reads in flight per thread
        float m[32];
#pragma unroll
        for (int e = 0; e < 32; e++) m[e] = Mtp[(size_t)(j0 + e) * KP];
#pragma unroll
        for (int e = 0; e < 32; e++) s = fmaf(m[e], sQv[64 * hj + j0 + e], s);
      }
      (hj ? sP : sY)[i] = s;
    }
    __syncthreads();
    float lt = 0.f, yy = 0.f;
    if (tid < KP) {
      const float y = sY[tid] + sP[tid];
      if (tid < k) a.Y[(size_t)row * k + tid] = y;
      yy = tid < k ? y * y : 0.f;
    }
    // 10. loss: x_j . y = v_j . q
    {
      const int j = tid >> 2, part = tid & 3;
      float s = 0.f;
      if (j < 32 * nrt) {
        const float* vr = sV + j * LD + 32 * part;
#pragma unroll 8
        for (int e = 0; e < 32; e++) s = fmaf(vr[e], sQv[32 * part + e], s);
      }
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      if (part == 0 && j < n) {
        const float dlt = 1.f - s;
        lt = sC[j] * dlt * dlt;
      }
    }
    const float lsum = wave_sum(lt), ysum = wave_sum(yy);
    if (lane == 0) wloss += (double)lsum + a.lambda_loss * (double)ysum;
    // shift the look-ahead
    row_c = row_n; p1_c = p1_n; n_c = n_n;
    row_n = row_nn; p1_n = p1_nn; n_n = n_nn;
    row_nn = row_n3;
    id_c = id_nx; c_c = c_nx;
  }
  __syncthreads();
  if (lane == 0) sRed[wv] = wloss;
  __syncthreads();
  if (tid == 0) a.loss_partials[loss_slot0 + blockIdx.x] = (sRed[0] + sRed[1]) + (sRed[2] + sRed[3]);
}

}  // namespace

bool chol_lr_supported(const AlsArgs& a, bool implicit) {
  return implicit && a.k > 96 && a.k <= 128 && a.k % 2 == 0 && !a.rhs_vals && !a.loss_tgt && !a.rhs_init &&
         (reinterpret_cast<uintptr_t>(a.X) & 7) == 0;
}

// rows: the n_rows rows of 1..kCholLrMax non-zeros (a suffix of the length-sorted order); M / Mt: 2 x 128 x 128 floats of
// scratch; flags: the device word launch_ne_stats leaves in stats[2] (some confidence < 1) -- the prep kernel ORs its own
// verdict into the same word.  Loss partials of its kCholLrGrid workgroups from loss_slot0 on.
hipError_t launch_als_chol_lr(const AlsArgs& a, const int32_t* rows, int n_rows, float* M, float* Mt, unsigned* flags,
                              int loss_slot0, hipStream_t s, hipEvent_t* ev_slot) {
  hipError_t err;
  if ((err = hipMemsetAsync(a.loss_partials + loss_slot0, 0, (size_t)kCholLrGrid * sizeof(double), s)) != hipSuccess)
    return err;
  if (n_rows <= 0) return hipSuccess;
  constexpr int KP = 128;
  auto prep = chol_lr_prep_kernel<KP>;
  const int prep_lds = 2 * KP * (KP + 1) * 4;
  if ((err = hipFuncSetAttribute(reinterpret_cast<const void*>(prep), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 prep_lds)) != hipSuccess)
    return err;
  hipLaunchKernelGGL(prep, dim3(1), dim3(256), prep_lds, s, a.XtX, a.k, M, Mt, flags);
  if ((err = hipGetLastError()) != hipSuccess) return err;
  auto kern = als_chol_lr_kernel<KP>;
  if ((err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)LrSmem<KP>::bytes)) != hipSuccess)
    return err;
  const int grid = n_rows < kCholLrGrid ? n_rows : kCholLrGrid;
  prof_note(ev_slot, reinterpret_cast<const void*>(kern));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LrSmem<KP>::bytes, s, a, rows, n_rows, M, Mt, flags, loss_slot0);
  return hipGetLastError();
}

}  // namespace rsparse_hip
