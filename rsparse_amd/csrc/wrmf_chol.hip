// Exact (Cholesky) half-iteration, register-blocked (gfx950, wave64).
//
// Replaces the solver == CHOLESKY branch of als_implicit<T> / als_explicit<T>
// (inst/include/wrmf_implicit.hpp:206-208,231,236; inst/include/wrmf_explicit.hpp:103-108):
//     lhs = XtX + X_nnz diag(c-1) X_nnz^T   (explicit: X_nnz X_nnz^T + lambda_use I),  rhs = X_nnz c,
//     y = solve(lhs, rhs)   -- Armadillo: LAPACK posv.
//
// One 256-thread workgroup per row.  The k x k system never touches LDS as a matrix: thread (I, K) of a
// 16 x 16 grid owns the BS x BS block (BS = KP/16) of the lower triangle in REGISTERS from the rank-one
// assembly through the factorisation:
//   assembly   chunks of 32 gathered vectors staged in LDS, acc[a][b] += x[I*BS+a] * c1 * x[K*BS+b]
//   factorise  right-looking blocked Cholesky over the 16 block columns: the diagonal thread factors its
//              block in registers, the panel threads do a BS x BS triangular solve, the trailing threads a
//              BS^3 update from two BS x BS blocks read from LDS (4 KB panel) -- 2 barriers per block column
//   solve      forward substitution with 16-lane DPP reductions along block rows, backward substitution
//              accumulating L^T y into an LDS vector (distinct addresses per thread, no atomics)
// The Gramian is read from L2 (64 KB per row, cached) instead of LDS, so a workgroup needs only ~25 KB of
// LDS and several rows are in flight per CU, overlapping one row's serial phases with another's.
#include "wrmf_internal.h"
#include "wrmf_device.h"

namespace rsparse_hip {
namespace {

using namespace dev;

__device__ __forceinline__ float row16_sum_c(float v) {
  v += dpp<0xB1>(v);
  v += dpp<0x4E>(v);
  v += dpp<0x141>(v);
  v += dpp<0x140>(v);
  return v;
}

template <int KP>
struct Chol2Smem {
  static constexpr int BS = KP / 16;
  static constexpr int TC = 32;
  static constexpr int LDT = KP + 4;
  static constexpr size_t tile_floats = (size_t)TC * LDT;          // gathered chunk
  static constexpr size_t panel_floats = (size_t)16 * BS * BS;     // L_IJ blocks of the current block column
  static constexpr size_t diag_floats = (size_t)BS * BS + BS;      // L_JJ + reciprocals of its diagonal
  static constexpr size_t vec_floats = (size_t)3 * KP + 2 * TC;    // rhs/z/y, S accumulator, spare, c, c1
  static constexpr size_t bytes = (tile_floats + panel_floats + diag_floats + vec_floats + 16) * 4 + 64;
};

// The 4 waves gather one 32-vector chunk: wave w fetches tile rows [8w, 8w+8).  VEC: all index loads, then
// all 16-byte vector loads of the wave are in flight together (2 dependent round trips per chunk).
template <int KP, bool VEC>
__device__ __forceinline__ void chol_gather_chunk(const AlsArgs& a, int base, int ccnt, float* sT, int wv, int lane) {
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

// 3 waves per SIMD (168 VGPRs): three rows per CU overlap each other's serial phases; measured 0.160 s ->
// 0.133 s per 1M users at k = 128 against 2 (a fourth changes nothing and spills)
#ifndef RSP_CHOL_MINW
#define RSP_CHOL_MINW 3
#endif
// LONG: the launch for rows of more than kCholLongLen non-zeros (a.chol_long_rows, longest first).  A row of n
// non-zeros summed one rank-one update after the other in fp32 drifts by ~n eps (8e-4 on the 5e5-non-zero item of
// the 10M x 1M configuration); here the running block is folded into a second register block every kFold chunks, so
// no partial sum sees more than 128 terms before it joins a sum of n/128 terms.  Twice the registers: 2 waves per SIMD.
template <int KP, bool IMPLICIT, bool VEC, bool LONG>
__global__ __launch_bounds__(256, LONG ? 2 : RSP_CHOL_MINW) void als_chol2_kernel(AlsArgs a, int loss_slot0) {
  using SM = Chol2Smem<KP>;
  constexpr int BS = SM::BS, TC = SM::TC, LDT = SM::LDT, NB = 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sT = reinterpret_cast<float*>(smem);
  float* sL = sT + SM::tile_floats;        // [NB][BS][BS]
  float* sD = sL + SM::panel_floats;       // [BS][BS] L_JJ, then [BS] 1/diag
  float* sV = sD + SM::diag_floats;        // [KP] rhs -> z -> y
  float* sS = sV + KP;                     // [KP] backward accumulator
  float* sC = sS + 2 * KP;                 // [TC] confidence / rating
  float* sC1 = sC + TC;                    // [TC] rank-one weight
  int* sFlag = reinterpret_cast<int*>(sC1 + TC);
  double* sLoss = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(sFlag + 2) + 7) & ~(uintptr_t)7);

  const int tid = threadIdx.x, lane = tid & 63, wv = rfl(tid >> 6);
  const int I = tid >> 4, K = tid & 15;  // block row / block column owned by this thread
  const bool lower = I >= K;
  const int k = a.k;
  for (int e = tid; e < (int)SM::tile_floats; e += 256) sT[e] = 0.f;
  if (tid == 0) *sFlag = 0;
  __syncthreads();
  double wloss = 0.0;

  // short rows belong to the low-rank kernel (wrmf_chol_lr.hip) unless it stood down (flag set on the device)
  const bool lr_on = !LONG && ((IMPLICIT && a.lr_flags && rfl((int)a.lr_flags[0]) == 0) || a.lrx);
  // the main launch walks its own ranges of the length-sorted order (a.chol_list; see AlsArgs) instead of every column:
  // on the bench matrix it owns 1.5 of the 10 million user rows, and every skipped column was a dependent load
  const bool listed = !LONG && a.chol_list != nullptr;
  const int n_tail = listed ? (lr_on ? a.n_cols - a.chol_empty_first : a.n_cols - a.chol_first - a.chol_n_main) : 0;
  const int tail0 = listed ? (lr_on ? a.chol_empty_first : a.chol_first + a.chol_n_main) : 0;
  const int n_iter = LONG ? a.n_chol_long : (listed ? a.chol_n_main + n_tail : a.n_cols);
  auto row_at = [&](const int it2) {
    if (LONG) return a.chol_long_rows[it2];
    if (!listed) return it2;
    return a.chol_list[it2 < a.chol_n_main ? a.chol_first + it2 : tail0 + (it2 - a.chol_n_main)];
  };
  // the row id runs two rows ahead of the solve, its pointers one ahead
  int row_c = 0, pa_c = 0, pb_c = 0, row_n = 0;
  if ((int)blockIdx.x < n_iter) {
    row_c = row_at(blockIdx.x);
    pa_c = a.col_ptrs[row_c];
    pb_c = a.col_ptrs[row_c + 1];
  }
  if ((int)(blockIdx.x + gridDim.x) < n_iter) row_n = row_at(blockIdx.x + gridDim.x);
  for (int it = blockIdx.x; it < n_iter; it += gridDim.x) {
    const int row = rfl(row_c);
    const int p1 = rfl(pa_c), p2 = rfl(pb_c);
    {
      int pa_n = 0, pb_n = 0, row_nn = 0;
      if (it + (int)gridDim.x < n_iter) {
        const int rn = rfl(row_n);
        pa_n = a.col_ptrs[rn];
        pb_n = a.col_ptrs[rn + 1];
      }
      if (it + 2 * (int)gridDim.x < n_iter) row_nn = row_at(it + 2 * gridDim.x);
      row_c = row_n; pa_c = pa_n; pb_c = pb_n; row_n = row_nn;
    }
    const int cnt = p2 - p1;
    if (!LONG && a.n_chol_long > 0 && cnt > kCholLongLen) continue;   // the LONG launch owns it
    if (lr_on && cnt >= 1 && cnt <= kCholLrMax) continue;
    if (!LONG && a.ne_chol && cnt > a.ne_chol_min) continue;   // assembled and solved by wrmf_ne.hip
    float* yrow = a.Y + (size_t)row * k;
    if (cnt <= 0 && !a.rhs_init) {
      for (int e = tid; e < k; e += 256) yrow[e] = 0.f;
      continue;
    }
    const float lam_use =
        IMPLICIT ? 0.f : (float)(a.lambda_loss * (a.dynamic_lambda ? (double)(float)cnt : 1.0));

    // ---------------- assembly ----------------
    // the block starts as the Gramian (implicit, from L2) or lambda_use I (explicit); identity on the padded
    // diagonal -- loaded first, so no second set of 64 registers is live next to the accumulators
    float acc[BS][BS];
#pragma unroll
    for (int x = 0; x < BS; x++)
#pragma unroll
      for (int y = 0; y < BS; y++) {
        const int rr = I * BS + x, cc = K * BS + y;
        float gv;
        if (rr >= k || cc >= k) gv = (rr == cc) ? 1.f : 0.f;
        else if (IMPLICIT) gv = lower ? a.XtX[(size_t)rr * k + cc] : 0.f;
        else gv = (rr == cc) ? lam_use : 0.f;
        acc[x][y] = gv;
      }
    float acc2[LONG ? BS : 1][LONG ? BS : 1];
    if constexpr (LONG) {
#pragma unroll
      for (int x = 0; x < BS; x++)
#pragma unroll
        for (int y = 0; y < BS; y++) { acc2[x][y] = acc[x][y]; acc[x][y] = 0.f; }
    }
    constexpr int kFold = 4;
    int fold = 0;
    float rhs = 0.f;  // thread tid < KP owns rhs[tid]
    for (int base = p1; base < p2; base += TC) {
      const int ccnt = min(TC, p2 - base);
      __syncthreads();  // previous chunk (and the previous row's vectors) fully consumed
      if (tid < ccnt) {
        const float cvv = a.vals[base + tid];
        sC[tid] = a.rhs_vals ? a.rhs_vals[base + tid] : cvv;   // coefficient in the right-hand side
        sC1[tid] = IMPLICIT ? cvv - 1.f : 1.f;
      }
      chol_gather_chunk<KP, VEC>(a, base, ccnt, sT, wv, lane);
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
      if constexpr (LONG) {
        if (++fold == kFold) {
          fold = 0;
#pragma unroll
          for (int x = 0; x < BS; x++)
#pragma unroll
            for (int y = 0; y < BS; y++) { acc2[x][y] += acc[x][y]; acc[x][y] = 0.f; }
        }
      }
    }
    if constexpr (LONG) {
#pragma unroll
      for (int x = 0; x < BS; x++)
#pragma unroll
        for (int y = 0; y < BS; y++) acc[x][y] += acc2[x][y];
    }
    __syncthreads();
    if (tid < KP) sV[tid] = rhs + ((a.rhs_init && tid < k) ? a.rhs_init[tid] : 0.f);
    if (tid < KP) sS[tid] = 0.f;

    // ---------------- blocked Cholesky, lower triangle ----------------
    float dinv[BS];  // diagonal threads: 1 / L_cc of their block (the substitutions multiply instead of dividing)
#pragma unroll
    for (int c = 0; c < BS; c++) dinv[c] = 1.f;
    for (int J = 0; J < NB; J++) {
      if (I == J && K == J) {  // factor the diagonal block in registers
        float inv[BS];
#pragma unroll
        for (int c = 0; c < BS; c++) {
          float d = acc[c][c];
#pragma unroll
          for (int m = 0; m < BS; m++)
            if (m < c) d = fmaf(-acc[c][m], acc[c][m], d);
          if (!(d > 0.f)) { *sFlag = 1; d = 1.f; }
          inv[c] = __frsqrt_rn(d);   // one transcendental instead of sqrt + divide on the serial chain
          acc[c][c] = d * inv[c];
          dinv[c] = inv[c];
#pragma unroll
          for (int r2 = 0; r2 < BS; r2++) {
            if (r2 > c) {
              float v = acc[r2][c];
#pragma unroll
              for (int m = 0; m < BS; m++)
                if (m < c) v = fmaf(-acc[r2][m], acc[c][m], v);
              acc[r2][c] = v * inv[c];
            }
          }
        }
#pragma unroll
        for (int x = 0; x < BS; x++) {
#pragma unroll
          for (int y = 0; y < BS; y++) sD[x * BS + y] = y <= x ? acc[x][y] : 0.f;
          sD[BS * BS + x] = inv[x];
        }
      }
      __syncthreads();
      if (K == J && I > J) {  // panel: L_IJ = A_IJ L_JJ^{-T}, one column of L_JJ at a time (BS live LDS values)
#pragma unroll
        for (int c = 0; c < BS; c++) {
          float dc[BS];
#pragma unroll
          for (int m = 0; m < BS; m++) dc[m] = m < c ? sD[c * BS + m] : 0.f;
          const float ic = sD[BS * BS + c];
#pragma unroll
          for (int x = 0; x < BS; x++) {
            float v = acc[x][c];
#pragma unroll
            for (int m = 0; m < BS; m++)
              if (m < c) v = fmaf(-acc[x][m], dc[m], v);
            acc[x][c] = v * ic;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int x = 0; x < BS; x++)
#pragma unroll
          for (int y = 0; y < BS; y++) sL[(I * BS + y) * BS + x] = acc[x][y];
      }
      __syncthreads();
      if (K > J && I >= K) {  // trailing update: A_IK -= L_IJ L_KJ^T, one rank-one step per panel column m
        const float* pi = sL + I * BS * BS;   // blocks are stored column-major: pi[m*BS + x] = L_IJ[x][m]
        const float* pk = sL + K * BS * BS;
#pragma unroll 2
        for (int m = 0; m < BS; m++) {
          float li[BS], lk[BS];
#pragma unroll
          for (int x = 0; x < BS; x++) {
            li[x] = pi[m * BS + x];
            lk[x] = pk[m * BS + x];
          }
#pragma unroll
          for (int x = 0; x < BS; x++)
#pragma unroll
            for (int y = 0; y < BS; y++) acc[x][y] = fmaf(-li[x], lk[y], acc[x][y]);
        }
      }
      // no barrier: the next diagonal thread only needs its own registers; sD / sL are rewritten only
      // after barriers that every thread reaches after finishing this update
    }

    // ---------------- forward substitution  L z = b  (z overwrites sV) ----------------
    for (int J = 0; J < NB; J++) {
      __syncthreads();  // z blocks < J published
      if (I == J) {
        float part[BS];
#pragma unroll
        for (int x = 0; x < BS; x++) part[x] = 0.f;
        if (K < J) {
#pragma unroll
          for (int y = 0; y < BS; y++) {
            const float zk = sV[K * BS + y];
#pragma unroll
            for (int x = 0; x < BS; x++) part[x] = fmaf(acc[x][y], zk, part[x]);
          }
        }
#pragma unroll
        for (int x = 0; x < BS; x++) part[x] = row16_sum_c(part[x]);  // the 16 threads of block row J are one DPP row
        if (K == J) {
          float z[BS];
#pragma unroll
          for (int c = 0; c < BS; c++) {
            float v = sV[J * BS + c] - part[c];
#pragma unroll
            for (int m = 0; m < BS; m++)
              if (m < c) v = fmaf(-acc[c][m], z[m], v);
            z[c] = v * dinv[c];
          }
#pragma unroll
          for (int c = 0; c < BS; c++) sV[J * BS + c] = z[c];
        }
      }
    }
    // ---------------- backward substitution  L^T y = z  (y overwrites sV) ----------------
    for (int J = NB - 1; J >= 0; J--) {
      __syncthreads();  // sS complete for block J, y blocks > J published
      if (I == J && K == J) {
        float y[BS];
#pragma unroll
        for (int c = BS - 1; c >= 0; c--) {
          float v = sV[J * BS + c] - sS[J * BS + c];
#pragma unroll
          for (int m = 0; m < BS; m++)
            if (m > c) v = fmaf(-acc[m][c], y[m], v);
          y[c] = v * dinv[c];
        }
#pragma unroll
        for (int c = 0; c < BS; c++) sV[J * BS + c] = y[c];
      }
      __syncthreads();
      if (I == J && K < J) {  // S_K += L_JK^T y_J   (one thread per K: no conflicts)
#pragma unroll
        for (int y2 = 0; y2 < BS; y2++) {
          float s = sS[K * BS + y2];
#pragma unroll
          for (int x = 0; x < BS; x++) s = fmaf(acc[x][y2], sV[J * BS + x], s);
          sS[K * BS + y2] = s;
        }
      }
    }
    __syncthreads();
    if (tid < k) yrow[tid] = sV[tid];
    // a non-positive pivot: the row goes to the general solver (wrmf_lu.hip), which also owns its loss term
    const bool rowbad = *sFlag != 0;
    if (rowbad) {
      __syncthreads();
      if (tid == 0) {
        const int pos = atomicAdd(a.fail_counter, 1);
        if (pos < a.fail_cap) a.fail_rows[pos] = row;
        else for (int e = 0; e < k; e++) yrow[e] = 0.f;   // no room in the list: unresolved, zeroed like a singular row (no NaN is left behind)
        *sFlag = 0;
      }
      continue;   // (uniform; the assembly of the next row has barriers before the flag can be set again)
    }

    // ---------------- loss row term (wrmf_implicit.hpp:259-261 / wrmf_explicit.hpp:131-132) ----------------
    // second pass over the row's chunks: t_j = y . x_j with the lane-per-non-zero dot of the CG kernels
    {
      float lacc = 0.f;
      for (int base = p1; base < p2; base += TC) {
        const int ccnt = min(TC, p2 - base);
        __syncthreads();  // tile free (and y published in sV on the first pass)
        if (tid < ccnt) {
          sC[tid] = a.vals[base + tid];
          sC1[tid] = a.loss_tgt ? a.loss_tgt[base + tid] : a.loss_tgt_const;
        }
        chol_gather_chunk<KP, VEC>(a, base, ccnt, sT, wv, lane);
        __syncthreads();
        if (wv == 0) {
          const float t = tile_dot<KP, TC>(sT, sV, lane);
          const int jl = lane % TC;
          const float cvv = sC[jl < ccnt ? jl : 0];
          const float d = IMPLICIT ? sC1[jl < ccnt ? jl : 0] - t : cvv - t;
          lacc += (jl < ccnt && lane < TC) ? (IMPLICIT ? cvv * d * d : d * d) : 0.f;
        }
      }
      if (wv == 0) {
        const float lpart = wave_sum(lacc);
        float xxp = 0.f;
        for (int e = lane; e < k; e += 64) xxp = fmaf(sV[e], sV[e], xxp);
        xxp = wave_sum(xxp);
        if (lane == 0)
          wloss += IMPLICIT ? (double)lpart + a.lambda_loss * (double)xxp : (double)(lpart + lam_use * xxp);
      }
    }
  }
  __syncthreads();
  if (lane == 0) sLoss[wv] = wloss;
  __syncthreads();
  if (tid == 0) {
    a.loss_partials[loss_slot0 + blockIdx.x] = (sLoss[0] + sLoss[1]) + (sLoss[2] + sLoss[3]);
  }
}

struct LongStream {
  hipStream_t st = nullptr;
  hipEvent_t fork = nullptr, done = nullptr;
  int device = -1;
  hipError_t ensure() {
    int dev = 0;
    hipError_t err = hipGetDevice(&dev);
    if (err != hipSuccess || dev == device) return err;
    device = dev;  // a process drives one GPU, so this happens once
    if ((err = hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) != hipSuccess) return err;
    if ((err = hipEventCreateWithFlags(&fork, hipEventDisableTiming)) != hipSuccess) return err;
    return hipEventCreateWithFlags(&done, hipEventDisableTiming);
  }
};
thread_local LongStream g_long_stream;

template <int KP, bool IMPLICIT, bool VEC>
hipError_t launch_chol2_t(const AlsArgs& a, hipStream_t s, hipEvent_t* ev) {
  using SM = Chol2Smem<KP>;
  hipError_t err;
  const int grid = (int)chol_loss_slots(a.n_cols);
  const int grid_long = a.ne_chol ? 0 : (a.n_chol_long < kCholLongGrid ? a.n_chol_long : kCholLongGrid);
  auto kc = als_chol2_kernel<KP, IMPLICIT, VEC, false>;
  auto kl = als_chol2_kernel<KP, IMPLICIT, VEC, true>;
  for (const void* f : {reinterpret_cast<const void*>(kc), reinterpret_cast<const void*>(kl)})
    if ((err = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SM::bytes)) != hipSuccess)
      return err;
  if ((err = hipMemsetAsync(a.loss_partials + grid, 0, (size_t)kCholLongGrid * sizeof(double), s)) != hipSuccess)
    return err;
  // per-kernel timing: ev[0] | low-rank kernel for the short rows (+ its one-workgroup prep) | ev[1] | k x k kernel(s) | ev[2]
  if (ev && (err = hipEventRecord(ev[0], s)) != hipSuccess) return err;
  if (a.lrx) {   // explicit feedback: the short rows in push-through form
    if ((err = launch_als_chol_lrx(a, a.lr_rows, a.n_lr, a.wave_stats, grid + kCholLongGrid, s, ev)) != hipSuccess) return err;
  } else if (a.lr_flags) {   // the short rows first (its prep kernel also settles lr_flags before the launches below read it)
    if ((err = launch_als_chol_lr(a, a.lr_rows, a.n_lr, a.lr_M, a.lr_M + 128 * 128, a.lr_flags, grid + kCholLongGrid, s,
                                  ev)) != hipSuccess)
      return err;
  } else if ((err = hipMemsetAsync(a.loss_partials + grid + kCholLongGrid, 0, (size_t)kCholLrGrid * sizeof(double), s)) !=
             hipSuccess) {
    return err;
  }
  if (ev && (err = hipEventRecord(ev[1], s)) != hipSuccess) return err;
  // The long rows are few and the longest of them decides when the half-iteration ends: their launch runs on a side
  // stream next to the main one (disjoint rows), forked from / joined to the caller's stream with events.  With
  // per-kernel timing (ev) both go back to back on the caller's stream.
  if (grid_long > 0) {
    hipStream_t ls = s;
    if (!ev) {
      if ((err = g_long_stream.ensure()) != hipSuccess) return err;
      ls = g_long_stream.st;
      if ((err = hipEventRecord(g_long_stream.fork, s)) != hipSuccess) return err;
      if ((err = hipStreamWaitEvent(ls, g_long_stream.fork, 0)) != hipSuccess) return err;
    }
    hipLaunchKernelGGL(kl, dim3(grid_long), dim3(256), SM::bytes, ls, a, grid);
    if ((err = hipGetLastError()) != hipSuccess) return err;
    if (!ev) {
      if ((err = hipEventRecord(g_long_stream.done, ls)) != hipSuccess) return err;
      if ((err = hipStreamWaitEvent(s, g_long_stream.done, 0)) != hipSuccess) return err;
    }
  }
  if (chol_wave_supported(a.k) && !a.lr_flags) {   // rank <= 64: one wave per row, the system in its registers
    if ((err = launch_als_chol_wave(a, IMPLICIT, grid, 0, s, ev ? ev + 1 : nullptr)) != hipSuccess) return err;
  } else {
    prof_note(ev ? ev + 1 : nullptr, reinterpret_cast<const void*>(kc));
    hipLaunchKernelGGL(kc, dim3(grid), dim3(256), SM::bytes, s, a, 0);
    if ((err = hipGetLastError()) != hipSuccess) return err;
  }
  if (ev && (err = hipEventRecord(ev[2], s)) != hipSuccess) return err;
  return hipSuccess;
}

}  // namespace

size_t chol2_loss_slots(int n_cols) { return chol_loss_slots(n_cols) + (size_t)kCholLongGrid + (size_t)kCholLrGrid; }

hipError_t launch_als_chol2(const AlsArgs& a, bool implicit, hipStream_t s, hipEvent_t* ev) {
  const int KP = padded_rank(a.k);
  const bool vec = (a.k % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.X) & 15) == 0);
#define RSP_DISPATCH(KPV)                                                                                  \
  if (KP == KPV) {                                                                                         \
    if (implicit) return vec ? launch_chol2_t<KPV, true, true>(a, s, ev) : launch_chol2_t<KPV, true, false>(a, s, ev); \
    return vec ? launch_chol2_t<KPV, false, true>(a, s, ev) : launch_chol2_t<KPV, false, false>(a, s, ev); \
  }
  RSP_DISPATCH(32)
  RSP_DISPATCH(64)
  RSP_DISPATCH(128)
#undef RSP_DISPATCH
  return hipErrorInvalidValue;
}

}  // namespace rsparse_hip
