// Exact (Cholesky) half-iteration at rank <= 64: one WAVE per row, the row's k x k system in the wave's registers
// (gfx950, wave64; round 4).
//
// Replaces, for these ranks, the main launch of wrmf_chol.hip -- the solver == CHOLESKY branch of als_implicit<T> /
// als_explicit<T> (inst/include/wrmf_implicit.hpp:207-208,231,236; wrmf_explicit.hpp:103-108):
//     lhs = XtX + X_nnz diag(c - 1) X_nnz^T   |   X_nnz X_nnz^T + lambda_use I,      rhs = X_nnz c (+ rhs_init),
//     Y_new = solve(lhs, rhs, fast + likely_sympd).
// The 256-thread kernel there spends its time waiting (SQ counters on config 5, rank 64: waves 21 % issuing, 69 % waiting:
// a dozen barriers per block column for a 64 x 64 system that one wave can hold).  Here lane l holds row l of lhs in KP
// registers from the assembly (every gathered vector is broadcast through 256 bytes of LDS: 16 reads and 64 FMAs per
// non-zero) through a right-looking LDL^T in which the pivot row reaches the lanes as a DPP row broadcast inside the FMA
// (the register solve of wrmf_chol_lr.hip, with a guarded pivot) to both substitutions: no barrier, no matrix in LDS, twelve
// rows in flight per CU.  A non-positive pivot sends the row to the general solver (wrmf_lu.hip), exactly as the k x k
// kernel does.  Rows: the ones the main launch of wrmf_chol.hip would take (everything the normal-equation launch and the
// two-level LONG launch do not own).
#include <utility>

#include "wrmf_internal.h"
#include "wrmf_device.h"

namespace rsparse_hip {
namespace {

using namespace dev;

template <class F, int... I>
__device__ __forceinline__ void cw_sfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void cw_sfor(F&& f) {
  cw_sfor_impl(f, std::make_integer_sequence<int, N>{});
}

template <int KP, bool IMPLICIT>
__global__ __launch_bounds__(256, 3) void als_chol_wave_kernel(AlsArgs a, int loss_slot0) {
  __shared__ __attribute__((aligned(16))) float sXv[4][KP];   // per wave: the vector being broadcast
  __shared__ double sLoss[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = rfl(tid >> 6);
  float* sX = sXv[wv];
  const int k = a.k;
  const bool on = lane < KP, lk = lane < k;
  const bool vec = (k % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.X) & 15) == 0);
  double wloss = 0.0;

  // the rows of the main launch as ranges of the length-sorted order (see als_chol2_kernel)
  const bool listed = a.chol_list != nullptr;
  const int tail0 = listed ? a.chol_first + a.chol_n_main : 0;
  const int n_tail = listed ? a.n_cols - tail0 : 0;
  const int n_iter = listed ? a.chol_n_main + n_tail : a.n_cols;
  auto row_at = [&](const int it2) {
    if (!listed) return it2;
    return (int)a.chol_list[it2 < a.chol_n_main ? a.chol_first + it2 : tail0 + (it2 - a.chol_n_main)];
  };

  for (int it = blockIdx.x * 4 + wv; it < n_iter; it += gridDim.x * 4) {
    const int row = rfl(row_at(it));
    const int p1 = rfl(a.col_ptrs[row]), p2 = rfl(a.col_ptrs[row + 1]);
    const int cnt = p2 - p1;
    if (a.n_chol_long > 0 && cnt > kCholLongLen) continue;   // the LONG launch owns it
    if (a.ne_chol && cnt > a.ne_chol_min) continue;          // assembled and solved by wrmf_ne.hip
    float* yrow = a.Y + (size_t)row * k;
    if (cnt <= 0 && !a.rhs_init) {  // empty column -> zeros (wrmf_implicit.hpp:281, wrmf_explicit.hpp:142)
      if (lk) yrow[lane] = 0.f;
      continue;
    }
    const float lam_use =
        IMPLICIT ? 0.f : (float)(a.lambda_loss * (a.dynamic_lambda ? (double)(float)cnt : 1.0));
    // the lane id as this row sees it (hipcc otherwise hoists the lane-dependent addresses and compares of the unrolled
    // loops out of the row loop and spills them)
    int ln = lane;
    asm volatile("" : "+v"(ln));

    // ---------------- assembly: lane l holds lhs(l, :) and rhs_l ----------------
    float r[KP];
    {
      const float* gcol = a.XtX + ln;
#pragma unroll
      for (int m = 0; m < KP; m++) {
        float gv;
        if (m >= k || !lk) gv = (m == ln) ? 1.f : 0.f;   // padded coordinates: identity, rhs 0
        else if (IMPLICIT) gv = gcol[(size_t)m * k];
        else gv = (m == ln) ? lam_use : 0.f;
        r[m] = gv;
      }
    }
    float u = 0.f;
    for (int base = p1; base < p2; base += 64) {
      const int ccnt = min(64, p2 - base);
      const int jl = min(lane, ccnt - 1);
      const int idj = a.row_idx[base + jl];
      const float cvj = a.vals[base + jl];
      const float rcj = a.rhs_vals ? a.rhs_vals[base + jl] : cvj;   // coefficient in the right-hand side
      constexpr int PF = 4;                                         // vectors in flight
      float xq[PF];
#pragma unroll
      for (int q = 0; q < PF; q++) {
        const int id = __builtin_amdgcn_readlane(idj, min(q, ccnt - 1));
        xq[q] = lk ? a.X[(size_t)id * k + ln] : 0.f;
      }
      for (int j0 = 0; j0 < ccnt; j0 += PF) {
        float xn[PF];
#pragma unroll
        for (int q = 0; q < PF; q++) {
          const int id = __builtin_amdgcn_readlane(idj, min(j0 + PF + q, ccnt - 1));
          xn[q] = lk ? a.X[(size_t)id * k + ln] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < PF; q++) {
          if (j0 + q < ccnt) {   // wave-uniform
            const float xv = xq[q];
            const float cv = readlane_f(cvj, j0 + q);
            const float f = (IMPLICIT ? cv - 1.f : 1.f) * xv;
            u = fmaf(readlane_f(rcj, j0 + q), xv, u);
            wave_sync();
            if (on) sX[ln] = xv;
            wave_sync();
#pragma unroll
            for (int m4 = 0; m4 < KP / 4; m4++) {
              const float4 b = *reinterpret_cast<const float4*>(sX + 4 * m4);   // broadcast
              r[4 * m4 + 0] = fmaf(b.x, f, r[4 * m4 + 0]);
              r[4 * m4 + 1] = fmaf(b.y, f, r[4 * m4 + 1]);
              r[4 * m4 + 2] = fmaf(b.z, f, r[4 * m4 + 2]);
              r[4 * m4 + 3] = fmaf(b.w, f, r[4 * m4 + 3]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < PF; q++) xq[q] = xn[q];
      }
    }
    if (a.rhs_init && lk) u += a.rhs_init[ln];

    // ---------------- LDL^T, lane i = row i (wrmf_chol_lr.hip's register solve with a guarded pivot) ----------------
    // Pivot step j: r[c] -= l_ij A(j, c) for the columns c > j; A(j, c) = A(c, j) is lane c of register r[j]: its rows of
    // 16 lanes are copied into every row of 16 once per pivot and the multiplier is a DPP row broadcast inside the FMA.  The
    // next pivot column is served first and by v_readlane, so that its chain starts before this pivot's other columns are done.
    // At the end lane i holds, left of the diagonal, row i of L D and, right of it, column i of L d_i: both substitutions
    // read only the lane's own registers and broadcasts; the forward substitution rides along.
    bool bad = false;
    float dinv = 1.f;
    float pj = readlane_f(r[0], 0);
    cw_sfor<KP>([&](auto jt) {
      constexpr int j = decltype(jt)::value;
      bad = bad || !(pj > 0.f);
      const float i0 = __builtin_amdgcn_rcpf(pj);
      const float inv = fmaf(fmaf(-pj, i0, 1.f), i0, i0);
      const float uj = readlane_f(u, j);
      if (ln == j) dinv = inv;
      const float lij = ln > j ? r[j] * inv : 0.f;   // L_ij; rows <= j are finished
      u = fmaf(-lij, uj, u);
      if constexpr (j + 1 < KP) {
        r[j + 1] = fmaf(-lij, readlane_f(r[j], j + 1), r[j + 1]);
        pj = readlane_f(r[j + 1], j + 1);
        if constexpr (j + 2 < KP) {
          float rep[4];
          dpp_ready(r[j]);
          rows_to_all<(KP > 32 ? 4 : 2)>(r[j], rep);
          dpp_ready(rep[0], rep[1], rep[2], rep[3]);
          cw_sfor<KP - j - 2>([&](auto ct) {
            constexpr int c = j + 2 + decltype(ct)::value;
            fnma_row_bcast<c % 16>(r[c], rep[c / 16], lij);
          });
        }
      }
    });
    // backward: z_c = (u_c - sum_{c' > c} d_c L[c'][c] z_c') / d_c, largest index first
    float acc = 0.f, z = 0.f;
#pragma unroll
    for (int c = KP - 1; c >= 0; c--) {
      if (ln == c) z = (u - acc) * dinv;
      const float zc = readlane_f(z, c);
      acc = fmaf(ln < c ? r[c] : 0.f, zc, acc);
    }
    if (bad) {   // wave-uniform: the general solver re-solves the row and owns its loss term (wrmf_lu.hip)
      int pos = 0;
      if (lane == 0) pos = atomicAdd(a.fail_counter, 1);
      pos = rfl(pos);
      if (pos < a.fail_cap) {
        if (lane == 0) a.fail_rows[pos] = row;
      } else if (lk) {
        yrow[ln] = 0.f;   // no room in the list: unresolved, zeroed like a singular row
      }
      continue;
    }
    if (lk) yrow[ln] = z;

    // ---------------- loss row term: lane j takes non-zero j of a chunk ----------------
    wave_sync();
    if (on) sX[ln] = z;
    wave_sync();
    float lacc = 0.f;
    for (int base = p1; base < p2; base += 64) {
      const int ccnt = min(64, p2 - base);
      const int jl = min(lane, ccnt - 1);
      const float* xr = a.X + (size_t)a.row_idx[base + jl] * k;
      const float cvv = a.vals[base + jl];
      const float tgt = a.loss_tgt ? a.loss_tgt[base + jl] : a.loss_tgt_const;
      float t0 = 0.f, t1 = 0.f;
      if (vec) {
        for (int m4 = 0; m4 < k / 4; m4++) {
          const float4 xv = *reinterpret_cast<const float4*>(xr + 4 * m4);
          const float4 b = *reinterpret_cast<const float4*>(sX + 4 * m4);
          t0 = fmaf(xv.x, b.x, t0);
          t1 = fmaf(xv.y, b.y, t1);
          t0 = fmaf(xv.z, b.z, t0);
          t1 = fmaf(xv.w, b.w, t1);
        }
      } else {
        for (int m = 0; m < k; m++) t0 = fmaf(xr[m], sX[m], t0);
      }
      const float tt = t0 + t1;
      const float d = IMPLICIT ? tgt - tt : cvv - tt;
      lacc += lane < ccnt ? (IMPLICIT ? cvv * d * d : d * d) : 0.f;
    }
    const float lpart = wave_sum(lacc);
    const float xxp = wave_sum(z * z);
    wloss += IMPLICIT ? (double)lpart + a.lambda_loss * (double)xxp : (double)(lpart + lam_use * xxp);
  }
  if (lane == 0) sLoss[wv] = wloss;
  __syncthreads();
  if (tid == 0) a.loss_partials[loss_slot0 + blockIdx.x] = (sLoss[0] + sLoss[1]) + (sLoss[2] + sLoss[3]);
}

}  // namespace

bool chol_wave_supported(int k) { return padded_rank(k) == 32 || padded_rank(k) == 64; }

// grid workgroups (4 rows each at a time); loss partials [loss_slot0, loss_slot0 + grid)
hipError_t launch_als_chol_wave(const AlsArgs& a, bool implicit, int grid, int loss_slot0, hipStream_t s, hipEvent_t* ev_slot) {
  const int KP = padded_rank(a.k);
#define RSP_CW(KPV, IMP)                                                                  \
  {                                                                                       \
    auto kern = als_chol_wave_kernel<KPV, IMP>;                                           \
    prof_note(ev_slot, reinterpret_cast<const void*>(kern));                              \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, s, a, loss_slot0);                 \
    return hipGetLastError();                                                             \
  }
  if (KP == 32) {
    if (implicit) RSP_CW(32, true) else RSP_CW(32, false)
  }
  if (KP == 64) {
    if (implicit) RSP_CW(64, true) else RSP_CW(64, false)
  }
#undef RSP_CW
  return hipErrorInvalidValue;
}

}  // namespace rsparse_hip
