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
#include <utility>

#include "wrmf_internal.h"
#include "wrmf_device.h"

namespace rsparse_hip {
namespace {

using namespace dev;

constexpr int kScdMaxIter = 10000;   // SCD_MAX_ITER, inst/include/wrmf.hpp:20
constexpr float kScdTol = 1e-4f;     // SCD_TOL, inst/include/wrmf.hpp:21
constexpr float kNnlsEps = 1e-16f;   // EPS, inst/include/nnls.hpp:8

template <int KP, bool GLHS = false>
struct NnlsSmem {
  // GLHS (end of round 6, rank 65..128): lhs lives in a global scratch of the workgroup (L2) and the tile holds 16 vectors, so that
  // the squared system, which the sweeps read, is all the LDS holds -- 75 KB instead of 148: TWO workgroups per CU.  A row's cost is
  // its descent, one wave's serial chain; what fills the CU meanwhile is another workgroup's.
  static constexpr int BS = KP / 16;
  static constexpr int TC = GLHS ? 16 : 32;
  static constexpr int LDT = KP + 4;
  static constexpr size_t tile_floats = (size_t)TC * LDT;
  static constexpr size_t mat_floats = (size_t)KP * KP;
  static constexpr size_t vec_floats = (size_t)3 * KP + 2 * TC;  // rhs, init/result, spare, c, c1
  static constexpr size_t bytes = (tile_floats + (GLHS ? 1 : 2) * mat_floats + vec_floats + 16) * 4 + 64;
};

template <int KP, bool VEC, int TC = 32>
__device__ __forceinline__ void nnls_gather_chunk(const AlsArgs& a, int base, int ccnt, float* sT, int wv, int lane) {
  constexpr int LDT = KP + 4, TCW = TC / 4;   // (TCW vectors of the chunk per wave)
  const int k = a.k;
  if constexpr (VEC) {
    constexpr int LPV = KP / 4, VPI = 64 / LPV, NQ = TCW / VPI;
    static_assert(NQ >= 1, "a wave's share of the chunk is at least one load instruction");
    const int c4 = lane % LPV, jo = lane / LPV;
    int ids[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) ids[q] = a.row_idx[base + min(TCW * wv + q * VPI + jo, ccnt - 1)];
    float4 v[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) v[q] = *reinterpret_cast<const float4*>(a.X + (size_t)ids[q] * k + min(c4 * 4, k - 4));
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int j = TCW * wv + q * VPI + jo;
      if (j < ccnt && c4 * 4 < k) *reinterpret_cast<float4*>(sT + j * LDT + c4 * 4) = v[q];
    }
  } else {
    for (int j = TCW * wv; j < min(TCW * wv + TCW, ccnt); j++) {
      const int id = rfl(a.row_idx[base + j]);
      const float* src = a.X + (size_t)id * k;
      for (int e = lane; e < k; e += 64) sT[j * LDT + e] = src[e];
    }
  }
}

template <int KP, bool IMPLICIT, bool VEC, bool GLHS = false>
__global__ __launch_bounds__(256) void als_nnls_kernel(AlsArgs a) {
  using SM = NnlsSmem<KP, GLHS>;
  constexpr int BS = SM::BS, TC = SM::TC, LDT = SM::LDT, NS = KP / 64;  // NS coordinates per lane of wave 0
  static_assert(KP % 64 == 0 || KP == 32, "rank padding");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sT = reinterpret_cast<float*>(smem);
  // lhs, full symmetric, row-major: in LDS, or (GLHS) this workgroup's KP x KP floats of a global scratch -- written and read by
  // this workgroup only, between its barriers
  float* sA = GLHS ? a.nnls_lhs + (size_t)blockIdx.x * SM::mat_floats : sT + SM::tile_floats;
  float* sB = GLHS ? sT + SM::tile_floats : sA + SM::mat_floats;    // XtX = lhs^2
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

  for (int it = blockIdx.x; it < a.n_cols; it += gridDim.x) {
    const int row = a.nnls_order ? rfl(a.nnls_order[it]) : it;   // (longest first: see AlsArgs)
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
      nnls_gather_chunk<KP, VEC, TC>(a, base, ccnt, sT, wv, lane);
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
      // mu / XtX_kk as the wave-per-row kernel forms it (end of round 6): the reciprocal refined once per row, a residual correction
      // per quotient -- the correctly rounded quotient in three vector instructions for every lane at once, instead of a full
      // division (thirty) on three values read lane by lane; and the moving coordinate's column of XtX is requested from LDS before
      // the step is known (one wave sweeps while the workgroup waits: the read's latency was exposed at every visit)
      float rdg[NS > 0 ? NS : 1];
#pragma unroll
      for (int s2 = 0; s2 < NSL; s2++) {
        const float r0 = __builtin_amdgcn_rcpf(dg[s2]);
        rdg[s2] = fmaf(fmaf(-dg[s2], r0, 1.f), r0, r0);
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
            const float* brow = sB + kk * KP;  // column kk = row kk (symmetric)
            float bcol[NS > 0 ? NS : 1];
#pragma unroll
            for (int s3 = 0; s3 < NSL; s3++) bcol[s3] = brow[min(lane + 64 * s3, KP - 1)];
            const float q0 = mu[s2] * rdg[s2];
            const float qv = fmaf(fmaf(-q0, dg[s2], mu[s2]), rdg[s2], q0);   // mu / dg, correctly rounded, every lane's
            const float nvv = fmaxf(h[s2] - qv, 0.f);
            const float old_v = readlane_f(h[s2], l);
            const float new_v = readlane_f(nvv, l);
            const float diff = new_v - old_v;
            const unsigned long long above = l >= 63 ? 0ull : (~0ull << (l + 1));
            if (diff != 0.f) {  // wave-uniform
              if (lane == l) h[s2] = new_v;
#pragma unroll
              for (int s3 = 0; s3 < NSL; s3++) {
                const int c = lane + 64 * s3;
                if (c < KP) mu[s3] = fmaf(diff, bcol[s3], mu[s3]);
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
        nnls_gather_chunk<KP, VEC, TC>(a, base, ccnt, sT, wv, lane);
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

// ---- rank <= 64: one WAVE per row, the squared system in the wave's registers (round 4) ----
// What a row costs is the descent -- 172 sweeps per row on the config-2 shape, each a serial chain over the coordinates --
// and in the 256-thread kernel above one wave walks it while three wait and the two k x k matrices in LDS allow three
// workgroups per CU: three chains per CU.  Here every wave owns a row from the gather to the loss, nothing is shared and
// there is no barrier: lane l holds column l of lhs while it is assembled (KP registers), then column l of XtX = lhs^T lhs
// (KP registers: the update mu += diff XtX[:, c] of a coordinate c reads register c, the sweep is unrolled over c), mu_l, h_l
// and XtX_ll; LDS (half of lhs at a time, 8.7 KB per wave) only carries the broadcasts of the assembly and of the square.
// Twelve chains per CU (168 registers: three waves per SIMD).
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>) in order: an unrolled loop whose index is a constant
// expression (the lane number of a v_writelane_b32 has to be one)
template <class F, int... Cs>
__device__ __forceinline__ void static_for_seq(F&& f, std::integer_sequence<int, Cs...>) {
  (f(std::integral_constant<int, Cs>{}), ...);
}
// h with lane C replaced by the (wave-uniform) value sval
template <int C>
__device__ __forceinline__ float writelane_c(float h, float sval) {
  asm("v_writelane_b32 %0, %1, %2" : "+v"(h) : "s"(sval), "n"(C));
  return h;
}

template <int KP>
struct NnlsWaveSmem {
  static constexpr int HR = KP / 2;          // columns of lhs staged at a time
  static constexpr int LDH = KP + 4;         // their stride (16-byte aligned broadcast reads, 4-way conflicts on the writes)
  static constexpr size_t wave_floats = (size_t)HR * LDH + 2 * KP;
  static constexpr size_t bytes = wave_floats * 4 + 64;   // (one wave per workgroup)
};

template <int KP, bool IMPLICIT>
__global__ __launch_bounds__(64, 3) void als_nnls_wave_kernel(AlsArgs a) {
  // One wave per WORKGROUP: a row's cost is its sweep count (median 114, a tenth of the rows above 390, the cap 10000), and a
  // workgroup of four waves held its slot until the slowest of its four rows was done.
  using SM = NnlsWaveSmem<KP>;
  constexpr int HR = SM::HR, LDH = SM::LDH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  float* sW = reinterpret_cast<float*>(smem);                                  // [HR][LDH] columns of lhs
  float* sX = sW + (size_t)HR * LDH;                                           // [KP] a vector to broadcast
  float* sR = sX + KP;                                                         // [KP] a second one
  const int k = a.k;
  const bool on = lane < KP, lk = lane < k;
  const bool vec = (k % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.X) & 15) == 0);
  const unsigned long long in_range = k >= 64 ? ~0ull : ((1ull << k) - 1ull);
  double wloss = 0.0;
#ifdef RSP_NNLS_PROF   // dev builds (tools/gpu_nnls_prof.sh): s_memtime ticks per phase, sweeps and coordinate visits, summed over the waves into a.ne_prof[32 ..]
  unsigned long long nt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nt0 = __builtin_amdgcn_s_memtime();
#define NNLS_TICK(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); nt[i] += t_ - nt0; nt0 = t_; }
#define NNLS_COUNT(i, n) nt[i] += (n);
#else
#define NNLS_TICK(i)
#define NNLS_COUNT(i, n)
#endif

  for (int it = blockIdx.x; it < a.n_cols; it += gridDim.x) {
    const int row = a.nnls_order ? rfl(a.nnls_order[it]) : it;
    const int p1 = rfl(a.col_ptrs[row]), p2 = rfl(a.col_ptrs[row + 1]);
    const int cnt = p2 - p1;
    float* yrow = a.Y + (size_t)row * k;
    if (cnt <= 0 && !a.rhs_init) {  // empty column -> zeros (wrmf_implicit.hpp:281, wrmf_explicit.hpp:142)
      if (lk) yrow[lane] = 0.f;
      continue;
    }
    const float lam_use =
        IMPLICIT ? 0.f : (float)(a.lambda_loss * (a.dynamic_lambda ? (double)(float)cnt : 1.0));
    // the lane id as this row sees it: hipcc otherwise hoists every lane-dependent address and compare of the unrolled
    // loops below out of the row loop (KP 64-bit pointers, KP masks) and spills them
    int ln = lane;
    asm volatile("" : "+v"(ln));
    NNLS_TICK(4)
    NNLS_COUNT(5, 1)

    // ---------------- assembly: lane l holds lhs(:, l) and rhs_l ----------------
    float acol[KP];
    {
      const float* gcol = a.XtX + ln;
#pragma unroll
      for (int m = 0; m < KP; m++) {
        float gv;
        if (m >= k || !lk) gv = (m == ln) ? 1.f : 0.f;   // padded coordinates: identity, rhs 0 -> they stay 0
        else if (IMPLICIT) gv = gcol[(size_t)m * k];
        else gv = (m == ln) ? lam_use : 0.f;
        acol[m] = gv;
      }
    }
    float rhs = 0.f;
    for (int base = p1; base < p2; base += 64) {
      const int ccnt = min(64, p2 - base);
      const int jl = min(lane, ccnt - 1);
      const int idj = a.row_idx[base + jl];
      const float cvj = a.vals[base + jl];
      const float rcj = a.rhs_vals ? a.rhs_vals[base + jl] : cvj;   // coefficient in the right-hand side
      constexpr int PF = 4;                                         // vectors in flight
      float xq[PF];
#pragma unroll
      for (int u = 0; u < PF; u++) {
        const int id = __builtin_amdgcn_readlane(idj, min(u, ccnt - 1));
        xq[u] = lk ? a.X[(size_t)id * k + ln] : 0.f;
      }
      for (int j0 = 0; j0 < ccnt; j0 += PF) {
        float xn[PF];
#pragma unroll
        for (int u = 0; u < PF; u++) {
          const int id = __builtin_amdgcn_readlane(idj, min(j0 + PF + u, ccnt - 1));
          xn[u] = lk ? a.X[(size_t)id * k + ln] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < PF; u++) {
          if (j0 + u < ccnt) {   // wave-uniform
            const float xv = xq[u];
            const float cv = readlane_f(cvj, j0 + u);
            const float f = (IMPLICIT ? cv - 1.f : 1.f) * xv;
            rhs = fmaf(readlane_f(rcj, j0 + u), xv, rhs);
            wave_sync();
            if (on) sX[ln] = xv;
            wave_sync();
#pragma unroll
            for (int m4 = 0; m4 < KP / 4; m4++) {
              const float4 b = *reinterpret_cast<const float4*>(sX + 4 * m4);   // broadcast
              acol[4 * m4 + 0] = fmaf(b.x, f, acol[4 * m4 + 0]);
              acol[4 * m4 + 1] = fmaf(b.y, f, acol[4 * m4 + 1]);
              acol[4 * m4 + 2] = fmaf(b.z, f, acol[4 * m4 + 2]);
              acol[4 * m4 + 3] = fmaf(b.w, f, acol[4 * m4 + 3]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < PF; u++) xq[u] = xn[u];
      }
    }
    if (a.rhs_init && lk) rhs += a.rhs_init[ln];
    float h = lk ? yrow[ln] : 0.f;  // init = current Y.col(i)  (wrmf_implicit.hpp:185)

    // ---------------- mu = XtX init - lhs^T rhs, first half: t_l = sum_m lhs(m, l) rhs_m ----------------
    NNLS_TICK(0)
    wave_sync();
    if (on) {
      sX[ln] = h;
      sR[ln] = rhs;
    }
    wave_sync();
    float mu = 0.f;
#pragma unroll
    for (int m4 = 0; m4 < KP / 4; m4++) {
      const float4 b = *reinterpret_cast<const float4*>(sR + 4 * m4);
      mu = fmaf(-acol[4 * m4 + 0], b.x, mu);
      mu = fmaf(-acol[4 * m4 + 1], b.y, mu);
      mu = fmaf(-acol[4 * m4 + 2], b.z, mu);
      mu = fmaf(-acol[4 * m4 + 3], b.w, mu);
    }

    // ---------------- XtX(r, l) = sum_m lhs(m, r) lhs(m, l)  (nnls.hpp:41-44), HR columns r at a time ----------------
    float m2[KP];
    float dg;
    {   // XtX(l, l): the lane's own column against itself, summed as the columns below are
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int m4 = 0; m4 < KP / 4; m4++) {
        s0 = fmaf(acol[4 * m4 + 0], acol[4 * m4 + 0], s0);
        s1 = fmaf(acol[4 * m4 + 1], acol[4 * m4 + 1], s1);
        s0 = fmaf(acol[4 * m4 + 2], acol[4 * m4 + 2], s0);
        s1 = fmaf(acol[4 * m4 + 3], acol[4 * m4 + 3], s1);
      }
      dg = (s0 + s1) + kNnlsEps;
    }
#pragma unroll
    for (int hh = 0; hh < 2; hh++) {
      wave_sync();
      if (ln >= hh * HR && ln < hh * HR + HR) {
        float* dst = sW + (size_t)(ln - hh * HR) * LDH;
#pragma unroll
        for (int m4 = 0; m4 < KP / 4; m4++)
          *reinterpret_cast<float4*>(dst + 4 * m4) = make_float4(acol[4 * m4], acol[4 * m4 + 1], acol[4 * m4 + 2], acol[4 * m4 + 3]);
      }
      wave_sync();
      float chain = 0.f;
#pragma unroll
      for (int r = 0; r < HR; r++) {
        // the column's address "depends" on the previous column's result: hipcc otherwise issues the broadcast reads of all
        // HR columns first and spills what they return (8 KB of scratch per lane at rank 64)
        int coff = r * LDH;   // (an integer offset, not the pointer: the loads stay LDS loads)
        asm volatile("" : "+v"(coff) : "v"(chain));
        const float* colr = sW + coff;
        float s0 = 0.f, s1 = 0.f;
        constexpr int MB = KP / 4 > 8 ? 8 : KP / 4;   // broadcast pieces in flight (32 registers: lhs and XtX hold 2 KP)
#pragma unroll
        for (int mb = 0; mb < KP / 4; mb += MB) {
          if (mb > 0) {
            asm volatile("" : "+v"(coff) : "v"(s0), "v"(s1));
            colr = sW + coff;
          }
#pragma unroll
          for (int m4 = mb; m4 < mb + MB; m4++) {
            const float4 b = *reinterpret_cast<const float4*>(colr + 4 * m4);
            s0 = fmaf(b.x, acol[4 * m4 + 0], s0);
            s1 = fmaf(b.y, acol[4 * m4 + 1], s1);
            s0 = fmaf(b.z, acol[4 * m4 + 2], s0);
            s1 = fmaf(b.w, acol[4 * m4 + 3], s1);
          }
        }
        const float s = s0 + s1;
        m2[hh * HR + r] = s + (ln == hh * HR + r ? kNnlsEps : 0.f);   // XtX.diag() += EPS (nnls.hpp:44)
        chain = s;
      }
    }
    // ... second half: + sum_c XtX(l, c) init_c   (the clobber keeps hipcc from reading the 16 broadcast pieces of init
    // before the square and carrying them -- 64 registers -- through it)
    asm volatile("" ::: "memory");
#pragma unroll
    for (int m4 = 0; m4 < KP / 4; m4++) {
      const float4 b = *reinterpret_cast<const float4*>(sX + 4 * m4);
      mu = fmaf(m2[4 * m4 + 0], b.x, mu);
      mu = fmaf(m2[4 * m4 + 1], b.y, mu);
      mu = fmaf(m2[4 * m4 + 2], b.z, mu);
      mu = fmaf(m2[4 * m4 + 3], b.w, mu);
    }

    // ---------------- scd_ls_update (nnls.hpp:10-34): only the coordinates that move are visited ----------------
    // Every lane keeps the step its own coordinate would take against the CURRENT mu -- nv = max(0, h - mu / XtX_ll),
    // df = nv - h: one vector division serves all coordinates, where a uniform one per visit would be the same instruction
    // count for one of them -- and the sweep walks the set bits of ballot(df != 0): the reference's sequence of updates
    // (an idle visit changes nothing, nnls.hpp:24).  After a move every mu has changed and the steps are recomputed.
    // The quotient is the correctly rounded one (reciprocal refined once per row, residual correction per quotient: the
    // sequence the compiler's own expansion uses, without its range scaling -- mu / XtX_ll is nowhere near the subnormals).
    // The stopping rule needs only  max step_err <= tol: a step far from the threshold is decided by two products, the exact
    // division fabs(diff) / (fabs(old) + EPS) runs only inside a +-0.1 % band around it.
    NNLS_TICK(1)
    float rdg;
    {
      const float r0 = __builtin_amdgcn_rcpf(dg);
      rdg = fmaf(fmaf(-dg, r0, 1.f), r0, r0);
    }
    auto step_of = [&](float& nv_, float& df_) {
      const float q0 = mu * rdg;
      const float q = fmaf(fmaf(-q0, dg, mu), rdg, q0);   // mu / dg, correctly rounded
      nv_ = fmaxf(h - q, 0.f);                            // the reference's `if (new_value < 0) new_value = 0`: one v_max_f32
      df_ = nv_ - h;
    };
    constexpr float kTolHi = 1.001e-4f, kTolLo = 0.999e-4f;
    for (int t = 0; t < kScdMaxIter; t++) {
      int moved_far = 0;   // some step of this sweep exceeded the tolerance (an int: a wave-uniform bool lives in a lane mask)
      float nv, df;
      step_of(nv, df);
      unsigned long long act = __ballot(df != 0.f) & in_range;
      NNLS_COUNT(6, 1)
      NNLS_COUNT(7, __builtin_popcountll(act))
      static_for_seq([&](auto c_tag) {
        constexpr int c = decltype(c_tag)::value;
        if (__builtin_expect((act >> c) & 1ull, 0)) {
          const float d_c = readlane_f(df, c);
          if (!moved_far) {   // step_err of this move against the tolerance
            const float den = fabsf(h) + kNnlsEps;
            const unsigned long long far = __ballot(fabsf(df) > kTolHi * den), near = __ballot(fabsf(df) >= kTolLo * den);
            if ((far >> c) & 1ull) moved_far = 1;
            else if ((near >> c) & 1ull) moved_far = __ballot(fabsf(d_c) / (fabsf(readlane_f(h, c)) + kNnlsEps) > kScdTol) != 0ull ? 1 : 0;
          }
          // lane c takes its step -- read and written BY LANE NUMBER: a select on `ln == c` makes 64 loop-invariant lane masks that
          // hipcc hoists out of the sweeps into 128 scalar registers, spills to a vector register and reads back two lanes per
          // visit (the 360 scalar spills of rounds 4-5)
          h = writelane_c<c>(h, readlane_f(nv, c));
          mu = fmaf(d_c, m2[c], mu);
          step_of(nv, df);
          act = __ballot(df != 0.f) & in_range;
        }
      }, std::make_integer_sequence<int, KP>{});
      if (!moved_far) break;
    }
    if (lk) yrow[ln] = h;
    NNLS_TICK(2)

    // ---------------- loss row term: lane j takes non-zero j of a chunk ----------------
    wave_sync();
    if (on) sX[ln] = h;
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
    const float xxp = wave_sum(h * h);
    wloss += IMPLICIT ? (double)lpart + a.lambda_loss * (double)xxp : (double)(lpart + lam_use * xxp);
    NNLS_TICK(3)
  }
  if (lane == 0) a.loss_partials[blockIdx.x] = wloss;
#ifdef RSP_NNLS_PROF
  if (a.ne_prof && lane == 0)
    for (int j = 0; j < 8; j++) atomicAdd(a.ne_prof + 32 + j, nt[j]);
#endif
}

template <int KP, bool IMPLICIT>
hipError_t launch_nnls_wave_t(const AlsArgs& a, hipStream_t s, hipEvent_t* ev) {
  using SM = NnlsWaveSmem<KP>;
  hipError_t err;
  const int grid = (int)chol_loss_slots(a.n_cols);
  auto kc = als_nnls_wave_kernel<KP, IMPLICIT>;
  if ((err = hipFuncSetAttribute(reinterpret_cast<const void*>(kc), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)SM::bytes)) != hipSuccess)
    return err;
  if (ev && (err = hipEventRecord(ev[0], s)) != hipSuccess) return err;
  prof_note(ev, reinterpret_cast<const void*>(kc));
  hipLaunchKernelGGL(kc, dim3(grid), dim3(64), SM::bytes, s, a);
  if ((err = hipGetLastError()) != hipSuccess) return err;
  if (ev) {
    if ((err = hipEventRecord(ev[1], s)) != hipSuccess) return err;
    if ((err = hipEventRecord(ev[2], s)) != hipSuccess) return err;
  }
  return hipSuccess;
}

template <int KP, bool IMPLICIT, bool VEC, bool GLHS = false>
hipError_t launch_nnls_t(const AlsArgs& a, hipStream_t s, hipEvent_t* ev) {
  if constexpr (KP == 128 && VEC && !GLHS) {   // two workgroups per CU where the caller brought the scratch for lhs
    if (a.nnls_lhs) return launch_nnls_t<KP, IMPLICIT, VEC, true>(a, s, ev);
  }
  using SM = NnlsSmem<KP, GLHS>;
  hipError_t err;
  const int grid = (int)chol_loss_slots(a.n_cols);
  auto kc = als_nnls_kernel<KP, IMPLICIT, VEC, GLHS>;
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
  // rank <= 64: one wave per row (see als_nnls_wave_kernel); rank 65..128: one workgroup per row, the squared system in LDS
  if (KP == 32) return implicit ? launch_nnls_wave_t<32, true>(a, s, ev) : launch_nnls_wave_t<32, false>(a, s, ev);
  if (KP == 64) return implicit ? launch_nnls_wave_t<64, true>(a, s, ev) : launch_nnls_wave_t<64, false>(a, s, ev);
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
