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
// so the per-row work is two small GEMMs on the matrix cores, an n x n LDL^T instead of a k x k Cholesky, and a handful of
// matrix-vector products.  The GEMMs (V' = X_nnz M and W W^T) run as fp16-term products like the long-row kernel's
// (wrmf_ne.hip): every fp32 operand is split exactly into two fp16 terms of a power-of-two multiple of itself and the three
// products of order < 2 are accumulated in fp32 by v_mfma_f32_32x32x16_f16 -- 2^-21 per product, 5x the rate of the fp32
// matrix instruction (round 3: V' 73 -> and S 56 -> ms of the user half of config 4; the fp32 version is the git history).  The loss needs no second gather:
// x_j . y = v_j . q with q = g - W^T S^-1 W g.  Any exact method satisfies the reference's `solve`; the parity bound
// (1e-4 against the fp64 oracle) is the same as for wrmf_chol.hip and is checked by the same tests.
//
// Round 3: the n x n systems are solved by one wave in registers with the pivot-row multipliers as DPP row broadcasts inside
// the FMAs, and rows of <= 16 / <= 32 non-zeros share a pass four / two at a time (see "the rows" below).
// Round 4: at rank 128 a pass belongs to ONE wave from the gather to the loss (als_chol_lrw_kernel, second half of this
// file): the same algebra regrouped so that nothing is reduced over the lanes that hold the slots, no barrier, no operand
// in LDS but a read-only copy of the terms of M^T -- 185 -> 55 ms on the user half of the bench matrix.  The workgroup kernel
// below keeps the ranks 98..126.  Explicit feedback (als_explicit<T>'s exact branch, wrmf_explicit.hpp:103-108) has the same
// form without a Gramian: als_chol_lrx_kernel, third part of this file, ranks 64 and 128.
//
// Needs every confidence >= 1 (D^1/2) and XtX positive definite: both are decided on the device (flags[0] != 0 ->
// this kernel returns at once and wrmf_chol.hip's kernel, which otherwise skips the short rows, takes them).
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "wrmf_internal.h"
#include "wrmf_device.h"

namespace rsparse_hip {
namespace {

using namespace dev;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// x (already scaled) -> fl16(x), fl16(x - fl16(x)) for a pair; the residual is exact in fp32
__device__ __forceinline__ void lr_split(const float x0, const float x1, unsigned& hi, unsigned& lo) {
  const f32x2 v = {x0, x1};
  const f16x2 h = __builtin_convertvector(v, f16x2);
  const f32x2 r = v - __builtin_convertvector(h, f32x2);
  const f16x2 l = __builtin_convertvector(r, f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}
// biased exponent e (1..253) of the power of two that brings `vmax` into [2^13, 2^14); 2^(e - 127) is the scale
__device__ __forceinline__ int lr_scale_exp(float vmax) {
  const int eb = (int)((__float_as_uint(vmax) >> 23) & 0xffu);
  return min(253, max(1, 267 - eb));
}
__device__ __forceinline__ float lr_pow2(int biased) { return __uint_as_float((unsigned)biased << 23); }

template <class F, int... I>
__device__ __forceinline__ void lr_sfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void lr_sfor(F&& f) {
  lr_sfor_impl(f, std::make_integer_sequence<int, N>{});
}

#ifndef RSP_LR_ABL
#define RSP_LR_ABL 0   // dev builds: timing-only ablations (1 no V' GEMM, 2 no S GEMM, 4 no LDL^T, 8 no substitution, 16 no y = M q, 32 no gather)
#endif
#ifndef RSP_LRW_ABL
#define RSP_LRW_ABL 0   // dev builds of the wave-per-pass kernel, timing only: 1 no V' GEMM, 2 no T GEMM, 4 no n x n solve, 8 no P GEMM
#endif
constexpr int kLrLd = 130;   // LDS row stride (floats) of V' (fp32)
constexpr int kLrLh = 136;   // ... (halves) of the fp16 terms of X_nnz and of W: 16-byte aligned rows for the operand reads
constexpr int kLrLs = 65;    // ... of the n x n system

// ---- XtX = L L^T, M = L^-T, Mt = M^T = L^-1: one workgroup, once per half-iteration ----------------------------------
// M and Mt are written KP x KP, zero padded.  flags[0] |= 1 when XtX is not positive definite.
// Also written: the two fp16 terms of Mt * 2^e (the B operand of V' = X_nnz M: M[kk][col] = Mt[col][kk], 8 consecutive kk
// per lane) into M16 = [2][KP][KP] halves, and e (biased exponent) into flags[1].
// MT16 (nullable; the wave-per-pass kernel's B operand of P = V' M^T) = [2][KP][KP] halves: MT16[c][16 chunk + q] = the terms
// of Mt[16 chunk + o][c] * 2^e with q = 8 h + e', o = 8 (e' / 4) + 4 h + (e' % 4): the 16 factor dimensions of a chunk in
// the order in which the two lane halves of an accumulator tile hold them (see als_chol_lrw_kernel).
template <int KP>
__global__ __launch_bounds__(256) void chol_lr_prep_kernel(const float* __restrict__ G, int k, _Float16* __restrict__ M16,
                                                           float* __restrict__ Mt, unsigned* __restrict__ flags,
                                                           _Float16* __restrict__ MT16) {
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
  float mx = 0.f;
  for (int e = tid; e < KP * KP; e += 256) {
    const int r = e / KP, c = e % KP;
    if (r < k && c <= r) mx = fmaxf(mx, fabsf(sZ[r * LD + c]));
  }
  __shared__ float smx[256];
  smx[tid] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) smx[tid] = fmaxf(smx[tid], smx[tid + o]);
    __syncthreads();
  }
  const int eM = lr_scale_exp(fmaxf(smx[0], 1e-30f));
  const float sM = lr_pow2(eM);
  if (tid == 0) flags[1] = (unsigned)eM;
  for (int e = tid; e < KP * KP; e += 256) {
    const int r = e / KP, c = e % KP;
    const bool in = r < k && c < k;
    const float v = (in && c <= r) ? sZ[r * LD + c] : 0.f;   // Mt = L^-1 (lower)
    Mt[e] = v;
    const _Float16 h = (_Float16)(v * sM);
    M16[e] = h;
    M16[KP * KP + e] = (_Float16)(v * sM - (float)h);
    if (MT16) {   // entry (kk = r, column c) of Mt -> row c, position of kk
      const int o = r & 15, q = 8 * ((o >> 2) & 1) + (o & 3) + 4 * (o >> 3);
      const int d = c * KP + (r & ~15) + q;
      MT16[d] = h;
      MT16[KP * KP + d] = (_Float16)(v * sM - (float)h);
    }
  }
}

// ---- the rows ----------------------------------------------------------------------------------------------------
// One PASS of a workgroup = 64 slots of non-zeros: one row of 33..64 non-zeros, two rows of 17..32 (32 slots each) or four
// rows of <= 16 (16 slots each; a slot beyond its row's length is a zero vector with confidence 0).  The per-pass fixed
// costs (a dozen barriers, the M fragments and the rows of M^T from L2, the gather round trip) are what a short row costs
// -- its arithmetic is nothing -- and half of the bench matrix's users have <= 24 non-zeros, so they share them.  The
// packed rows' S = I + W W^T is block diagonal (the cross blocks are dropped when it is written), i.e. independent systems
// living in disjoint groups of 16 or 32 lanes, and their LDL^T runs for all of them at once on lane-group-local DPP
// broadcasts (no v_readlane at all).
template <int KP>
struct LrSmem {
  static constexpr int NP = 64, RMAX = 4;
  static constexpr size_t x_floats = (size_t)NP * kLrLh;   // fp16 terms of X_nnz ([2][NP][kLrLh] halves), then of W, then S (NP x kLrLs floats)
  static constexpr size_t v_floats = (size_t)NP * kLrLd;   // V'
  static constexpr size_t vec_floats = 4 * NP + 4 * RMAX * KP + 64;
  static constexpr size_t bytes = (x_floats + v_floats + vec_floats) * 4 + 64;
};

template <int E>
__device__ __forceinline__ float lr_row_bcast(const float v) {   // lane E of this lane's row of 16 lanes
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + E, 0xf, 0xf, true));
}

template <int KP>
__global__ __launch_bounds__(256, 2) void als_chol_lr_kernel(AlsArgs a, const int32_t* __restrict__ rows, int n_rows,
                                                             int n64, int n32, const _Float16* __restrict__ M16,
                                                             const float* __restrict__ Mt,
                                                             const unsigned* __restrict__ flags, int loss_slot0) {
  using SM = LrSmem<KP>;
  constexpr int NP = SM::NP, LD = kLrLd, LS = kLrLs, LH = kLrLh;
  static_assert(KP == 128, "written for rank 97..128");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sX = reinterpret_cast<float*>(smem);
  float* sS = sX;                        // alias: X_nnz is dead once V' exists
  _Float16* sXh = reinterpret_cast<_Float16*>(sX);   // [NP][LH] leading fp16 terms of X_nnz * 2^ex, later of W * 2^ew
  _Float16* sXl = sXh + NP * LH;                     // [NP][LH] second terms
  float* sV = sX + SM::x_floats;
  float* sC = sV + SM::v_floats;         // [NP] confidences
  float* sQ = sC + NP;                   // [NP] sqrt(c - 1)
  float* sH = sQ + NP;                   // [NP] h = W g, then sqrt(c - 1) z
  float* sT = sH + NP;                   // [NP] spare
  float* sGv = sT + NP;                  // [4][KP] g of the pass's rows
  float* sQv = sGv + 4 * KP;             // [4][KP] q
  float* sY = sQv + 4 * KP;              // [4][KP] y (two partial halves are added through sP)
  float* sP = sY + 4 * KP;               // [4][KP] second half of y
  double* sRed = reinterpret_cast<double*>(sP + 4 * KP);   // [8]
  const int tid = threadIdx.x, lane = tid & 63, wv = rfl(tid >> 6);
  const int col = lane & 31, half = lane >> 5;
  const int k = a.k;
  if (flags[0] != 0) return;   // some confidence < 1 or XtX not positive definite: wrmf_chol.hip takes these rows
  // operand scales: max |X| from the statistics block in front of the flags (launch_ne_stats: flags = stats + 2), the
  // exponent of Mt from the prep kernel
  const int ex = lr_scale_exp(fmaxf(__uint_as_float(flags[-2]), 1e-30f)), eM = (int)flags[1];
  const float sx = lr_pow2(ex), inv_xm = lr_pow2(254 - ex) * lr_pow2(254 - eM);   // 2^-(ex - 127) * 2^-(eM - 127)

  // the passes: [0, P64) one row each, [P64, P64 + P32) two rows, then four; the list is longest first
  const int n16 = n_rows - n64 - n32;
  const int P64 = n64, P32 = (n32 + 1) >> 1, P16 = (n16 + 3) >> 2, n_pass = P64 + P32 + P16;
  // slot -> (row of the pass, non-zero of the row) for pass pp: lsh = log2(slots per row); li = list index of this lane's row
  struct PassGeo { int lsh, li, lim; };
  auto geo = [&](const int pp) {
    PassGeo g;
    if (pp < P64) { g.lsh = 6; g.li = pp; g.lim = n64; }
    else if (pp < P64 + P32) { g.lsh = 5; g.li = n64 + 2 * (pp - P64) + (lane >> 5); g.lim = n64 + n32; }
    else { g.lsh = 4; g.li = n64 + n32 + 4 * (pp - P64 - P32) + (lane >> 4); g.lim = n_rows; }
    if (pp >= n_pass) g.lim = 0;
    return g;
  };

  double wloss = 0.0;
  // Row metadata runs ahead of the solves so that the gather of a pass is ONE memory round trip (it was a chain of four:
  // row id -> pointers -> index -> vector, and that per vector): the row ids three passes ahead, their pointers two ahead,
  // the indices and confidences (lane j holds slot j) one ahead.  All per lane: lane j works for the row of slot j.
  const int G = gridDim.x;
  int it = blockIdx.x;
  int rid_c = -1, n_c = 0, rid_n = -1, p1_n = 0, n_n = 0, rid_nn = -1;
  int id_c = 0;
  float c_c = 1.f;
  {
    const PassGeo g0 = geo(it), g1 = geo(it + G), g2 = geo(it + 2 * G);
    int p1_c = 0;
    if (g0.li < g0.lim) {
      rid_c = rows[g0.li];
      p1_c = a.col_ptrs[rid_c];
      n_c = a.col_ptrs[rid_c + 1] - p1_c;
    }
    if (g1.li < g1.lim) {
      rid_n = rows[g1.li];
      p1_n = a.col_ptrs[rid_n];
      n_n = a.col_ptrs[rid_n + 1] - p1_n;
    }
    if (g2.li < g2.lim) rid_nn = rows[g2.li];
    const int nz = lane & ((1 << g0.lsh) - 1);
    if (nz < n_c) {
      id_c = a.row_idx[p1_c + nz];
      c_c = a.vals[p1_c + nz];
    }
  }
  for (; it < n_pass; it += G) {
    const int lsh = it < P64 ? 6 : (it < P64 + P32 ? 5 : 4);   // log2 of the slots per row
    const int nz_c = lane & ((1 << lsh) - 1);
    const bool valid = nz_c < n_c;
    const unsigned long long vmask = __ballot(valid);
    const int nrt = (vmask >> 32) ? 2 : 1;   // 32-slot tiles in use
    __syncthreads();                   // the previous pass's buffers are free
    // requests for the passes to come (consumed at the bottom of this iteration)
    int id_nx = 0, p1_nn = 0, n_nn = 0, rid_n3 = -1;
    float c_nx = 1.f;
    {
      const PassGeo g1 = geo(it + G), g3 = geo(it + 3 * G);
      const int nz1 = lane & ((1 << g1.lsh) - 1);
      if (nz1 < n_n) {
        id_nx = a.row_idx[p1_n + nz1];
        c_nx = a.vals[p1_n + nz1];
      }
      if (rid_nn >= 0) {
        p1_nn = a.col_ptrs[rid_nn];
        n_nn = a.col_ptrs[rid_nn + 1] - p1_nn;
      }
      if (g3.li < g3.lim) rid_n3 = rows[g3.li];
    }
    // 1. gather: wave w takes slots w, w + 4, ...; a lane copies 2 floats of each.  All of a wave's loads are issued
    // before the first LDS store (empty slots are stored as zeros).
    {
      float2 v[16];
      const bool on = !(RSP_LR_ABL & 32);
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int j = wv + 4 * u;
        const int id = __builtin_amdgcn_readlane(id_c, j);
        v[u] = (on && ((vmask >> j) & 1) && 2 * lane < k) ? *reinterpret_cast<const float2*>(a.X + (size_t)id * k + 2 * lane) : float2{0.f, 0.f};
      }
      if (nrt == 2) {
#pragma unroll
        for (int u = 8; u < 16; u++) {
          const int j = wv + 4 * u;
          const int id = __builtin_amdgcn_readlane(id_c, j);
          v[u] = (on && ((vmask >> j) & 1) && 2 * lane < k) ? *reinterpret_cast<const float2*>(a.X + (size_t)id * k + 2 * lane) : float2{0.f, 0.f};
        }
      }
      if (wv == 0) {
        const float c = on ? c_c : 1.f;
        sC[lane] = valid ? c : 0.f;
        sQ[lane] = valid ? sqrtf(fmaxf(c - 1.f, 0.f)) : 0.f;
      }
      auto put = [&](const int u) {
        const int j = wv + 4 * u;
        unsigned hi = 0u, lo = 0u;
        if ((vmask >> j) & 1) lr_split(v[u].x * sx, v[u].y * sx, hi, lo);
        *reinterpret_cast<unsigned*>(sXh + j * LH + 2 * lane) = hi;
        *reinterpret_cast<unsigned*>(sXl + j * LH + 2 * lane) = lo;
      };
#pragma unroll
      for (int u = 0; u < 8; u++) put(u);
      if (nrt == 2) {
#pragma unroll
        for (int u = 8; u < 16; u++) put(u);
      }
    }
    __syncthreads();
    // 2. V' = X_nnz M on the matrix cores: this wave's 32 columns, all row tiles, 16 factor dimensions per instruction.
    // B fragments: lane (n = col, kg = half) holds Mt16[32 wv + col][16 ch + 8 kg .. + 7] = M[those kk][that column]; M is
    // upper triangular, so the chunks beyond 2 wv + 1 are zero and skipped.  (The pointer is made opaque so that the loads
    // stay inside the row loop -- hoisted, they pin 64 registers for the whole launch.)
    float wmax = 0.f;   // max |W| of this wave's part (for the scale of the second product)
    if (!(RSP_LR_ABL & 1)) {
      const _Float16* Mp = M16 + (size_t)(32 * wv + col) * KP + 8 * half;
      asm volatile("" : "+v"(Mp));
      f16x8 bh[8], bl[8];
#pragma unroll
      for (int ch = 0; ch < 8; ch++)
        if (ch <= 2 * wv + 1) {
          bh[ch] = *reinterpret_cast<const f16x8*>(Mp + 16 * ch);
          bl[ch] = *reinterpret_cast<const f16x8*>(Mp + KP * KP + 16 * ch);
        }
      for (int rt = 0; rt < nrt; rt++) {
        f32x16 acc, acc2;
#pragma unroll
        for (int e = 0; e < 16; e++) acc[e] = acc2[e] = 0.f;
        const _Float16* xa = sXh + (32 * rt + col) * LH + 8 * half;
#pragma unroll
        for (int ch = 0; ch < 8; ch++)
          if (ch <= 2 * wv + 1) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(xa + 16 * ch);
            const f16x8 al = *reinterpret_cast<const f16x8*>(xa + NP * LH + 16 * ch);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[ch], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[ch], acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[ch], acc2, 0, 0, 0);
          }
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int i = 32 * rt + (e & 3) + 8 * (e >> 2) + 4 * half;
          const float vv = (acc[e] + acc2[e]) * inv_xm;
          sV[i * LD + 32 * wv + col] = vv;
          wmax = fmaxf(wmax, fabsf(vv) * sQ[i]);
        }
      }
    }
    for (int o = 32; o > 0; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o));
    if (lane == 0) sT[wv] = wmax;
    __syncthreads();   // V' complete, every wave is done with the fp16 terms of X_nnz
    // 2b. the fp16 terms of W = D^1/2 V' * 2^ew over the X_nnz terms (thread t: slot t / 4, 32 columns)
    const int ew = lr_scale_exp(fmaxf(fmaxf(fmaxf(sT[0], sT[1]), fmaxf(sT[2], sT[3])), 1e-30f));
    if (!(RSP_LR_ABL & 2)) {
      const int j = tid >> 2, part = tid & 3;
      if (j < 32 * nrt) {
        const float sw = lr_pow2(ew) * sQ[j];
        const float* vr = sV + j * LD + 32 * part;
#pragma unroll
        for (int e = 0; e < 32; e += 2) {
          unsigned hi, lo;
          lr_split(vr[e] * sw, vr[e + 1] * sw, hi, lo);
          *reinterpret_cast<unsigned*>(sXh + j * LH + 32 * part + e) = hi;
          *reinterpret_cast<unsigned*>(sXl + j * LH + 32 * part + e) = lo;
        }
      }
    }
    // 3. g_r = V'_r^T c_r for the rows r of the pass: thread (t, hh) takes the rows r = hh, hh + 2
    {
      const int t = tid & (KP - 1), hh = tid >> 7, sl = 1 << lsh;
      for (int r = hh; r < (64 >> lsh); r += 2) {
        // (eight slots per trip, their reads issued together: one read-wait-FMA per slot made these two loops -- this one and
        //  step 8 -- a fifth of the pass)
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        const int j0 = r << lsh, j1 = min(j0 + sl, 32 * nrt);   // (multiples of 16)
        for (int j = j0; j < j1; j += 8) {
          const float4 ca = *reinterpret_cast<const float4*>(sC + j), cb = *reinterpret_cast<const float4*>(sC + j + 4);
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; e++) v[e] = sV[(j + e) * LD + t];
          s0 = fmaf(ca.x, v[0], s0); s1 = fmaf(ca.y, v[1], s1); s2 = fmaf(ca.z, v[2], s2); s3 = fmaf(ca.w, v[3], s3);
          s0 = fmaf(cb.x, v[4], s0); s1 = fmaf(cb.y, v[5], s1); s2 = fmaf(cb.z, v[6], s2); s3 = fmaf(cb.w, v[7], s3);
        }
        sGv[r * KP + t] = (s0 + s1) + (s2 + s3);
      }
    }
    __syncthreads();
    // 4. h = D^1/2 V' g  (4 threads per slot)
    {
      const int j = tid >> 2, part = tid & 3;
      float s = 0.f;
      if (j < 32 * nrt) {
        const float* vr = sV + j * LD + 32 * part;
        const float* gr = sGv + (j >> lsh) * KP + 32 * part;
#pragma unroll 8
        for (int e = 0; e < 32; e++) s = fmaf(vr[e], gr[e], s);
      }
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      if (part == 0 && j < NP) sH[j] = s * sQ[j];
    }
    // 5. S = I + W W^T (lower tiles) on the matrix cores from the fp16 terms of W, then -> sS (over those terms); entries
    // between slots of different rows are dropped (the off-diagonal tile is such entries only unless the pass is one row)
    __syncthreads();   // the terms of W are complete (and g, h above have been formed from V')
    {
      const int rt = wv == 0 ? 0 : 1, ct = wv == 2 ? 1 : 0;
      const bool mine = !(RSP_LR_ABL & 2) && wv < (nrt == 1 ? 1 : 3) && (lsh == 6 || wv != 1);
      f32x16 acc, acc2;
#pragma unroll
      for (int e = 0; e < 16; e++) acc[e] = acc2[e] = 0.f;
      if (mine) {
        const _Float16* wa = sXh + (32 * rt + col) * LH + 8 * half;
        const _Float16* wb = sXh + (32 * ct + col) * LH + 8 * half;
#pragma unroll
        for (int ch = 0; ch < 8; ch++) {
          const f16x8 ah = *reinterpret_cast<const f16x8*>(wa + 16 * ch), al = *reinterpret_cast<const f16x8*>(wa + NP * LH + 16 * ch);
          const f16x8 bh2 = *reinterpret_cast<const f16x8*>(wb + 16 * ch), bl2 = *reinterpret_cast<const f16x8*>(wb + NP * LH + 16 * ch);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh2, acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl2, acc2, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh2, acc2, 0, 0, 0);
        }
      }
      __syncthreads();   // every wave has read its operands: the region becomes S
      const float inv_w2 = lr_pow2(254 - ew) * lr_pow2(254 - ew);
      if (mine) {
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int i = 32 * rt + (e & 3) + 8 * (e >> 2) + 4 * half, c2 = 32 * ct + col;
          const bool same = (i >> lsh) == (c2 >> lsh);
          sS[i * LS + c2] = (i == c2 ? 1.f : 0.f) + (same ? (acc[e] + acc2[e]) * inv_w2 : 0.f);
        }
      }
    }
    __syncthreads();
    // 6 + 7. z = S^-1 h by ONE wave, in registers, lane i = slot i.  Right-looking LDL^T: the Schur complements stay
    // symmetric, so the pivot row's entry for column c is lane c of the pivot COLUMN's register, and at the end lane i holds,
    // left of the diagonal, column values frozen at their pivot steps (row i of L times D) and, right of it, its own pivot row
    // (column i of L times d_i) -- both triangular solves read nothing but the lane's own registers and broadcasts.  The
    // forward substitution rides along with the elimination.  No barriers, no LDS traffic after the system is loaded.
    if (!(RSP_LR_ABL & 4) && wv == ((blockIdx.x + (it / G)) & 3)) {
      int i = lane;
      asm volatile("" : "+v"(i));   // laundered per pass: hipcc otherwise hoists the 64 load addresses below out of the loop and spills them
      // one row of 33..64 non-zeros: the whole wave is one system
      auto solve = [&](auto np_tag) {
        constexpr int NS = decltype(np_tag)::value;
        const int n = rfl(n_c);
        float r[NS];
#pragma unroll
        for (int c = 0; c < NS; c++) r[c] = (i < n && c < n) ? sS[(i >= c ? i * LS + c : c * LS + i)] : (i == c ? 1.f : 0.f);
        float u = i < n ? sH[i] : 0.f;
        float dinv = 1.f;   // 1 / d_i, set when row i is the pivot row
        // Pivot step j: r[c] -= l_ij S[j][c] for the columns c > j.  S[j][c] = S[c][j] is lane c of register r[j]: its rows
        // of 16 lanes are copied into every row once per pivot (rows_to_all) and the multiplier is then a DPP row broadcast
        // inside the FMA -- one 4.8-cycle instruction per entry instead of v_readlane + v_fma (12.7).  Column j + 1, the next
        // pivot column, is served first and by v_readlane, so that the next pivot's chain (broadcast, reciprocal, scale) starts
        // before this pivot's other columns are done.
        float pj = readlane_f(r[0], 0);
        lr_sfor<NS>([&](auto jt) {
          constexpr int j = decltype(jt)::value;
          const float inv = __builtin_amdgcn_rcpf(pj);   // (>= 1: S = I + W W^T; 1 ulp)
          const float uj = readlane_f(u, j);
          if (i == j) dinv = inv;
          const float lij = i > j ? r[j] * inv : 0.f;   // L_ij; rows <= j are finished
          u = fmaf(-lij, uj, u);
          if constexpr (j + 1 < NS) {
            r[j + 1] = fmaf(-lij, readlane_f(r[j], j + 1), r[j + 1]);
            pj = readlane_f(r[j + 1], j + 1);
            if constexpr (j + 2 < NS) {
              float rep[4];
              dpp_ready(r[j]);
              rows_to_all<4>(r[j], rep);
              dpp_ready(rep[0], rep[1], rep[2], rep[3]);
              lr_sfor<NS - j - 2>([&](auto ct) {
                constexpr int c = j + 2 + decltype(ct)::value;
                fnma_row_bcast<c % 16>(r[c], rep[c / 16], lij);
              });
            }
          }
        });
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
      // packed rows: 64 / SL independent SL x SL systems, one per group of SL lanes; register t = column t OF THE GROUP.
      // Every broadcast is local to the group: a DPP row broadcast (SL = 16: the group is one row of 16 lanes), or one
      // v_permlane16_swap that lays the group's two rows side by side, then the row broadcast (SL = 32)
      auto solve_packed = [&](auto sl_tag) {
        constexpr int SL = decltype(sl_tag)::value;
        const int il = i & (SL - 1), base = i - il;
        const bool live = i < 32 * nrt;   // (the second tile of slots is not even computed when it is empty)
        float r[SL];
#pragma unroll
        for (int t = 0; t < SL; t++) r[t] = live ? sS[(il >= t ? i * LS + base + t : (base + t) * LS + i)] : (il == t ? 1.f : 0.f);
        float u = live ? sH[i] : 0.f;
        float dinv = 1.f;
        // lane t of the group, of the value v held one per lane
        auto grp = [&](const float v, float (&rep)[2]) {
          if constexpr (SL == 32) {
            const unsigned uu = __float_as_uint(v);
            const auto sw = __builtin_amdgcn_permlane16_swap(uu, uu, false, false);   // rows (0, 0, 2, 2) and (1, 1, 3, 3)
            rep[0] = __uint_as_float(sw[0]);
            rep[1] = __uint_as_float(sw[1]);
          } else {
            rep[0] = rep[1] = v;
          }
        };
        lr_sfor<SL>([&](auto tt) {
          constexpr int t = decltype(tt)::value;
          float rep[2], ur[2];
          dpp_ready(r[t], u);   // (both were last written by the FMAs of the step before, which hipcc cannot see into)
          grp(r[t], rep);
          grp(u, ur);
          dpp_ready(rep[0], rep[1], ur[0], ur[1]);
          const float pv = lr_row_bcast<t % 16>(rep[t / 16]);
          const float inv = __builtin_amdgcn_rcpf(pv);
          if (il == t) dinv = inv;
          const float lij = il > t ? r[t] * inv : 0.f;
          fnma_row_bcast<t % 16>(u, ur[t / 16], lij);
          lr_sfor<SL - t - 1>([&](auto ct) {
            constexpr int c = t + 1 + decltype(ct)::value;
            fnma_row_bcast<c % 16>(r[c], rep[c / 16], lij);
          });
        });
        float acc = 0.f, z = 0.f;
        lr_sfor<SL>([&](auto tt) {
          constexpr int c = SL - 1 - decltype(tt)::value;
          if (il == c) z = (u - acc) * dinv;
          float zr[2];
          grp(z, zr);
          dpp_ready(zr[0], zr[1]);
          const float m = il < c ? -r[c] : 0.f;
          fnma_row_bcast<c % 16>(acc, zr[c / 16], m);   // acc += r[c] z_c
        });
        sH[i] = live ? z * sQ[i] : 0.f;   // D^1/2 z (empty slots: sQ = 0)
      };
      if (!(RSP_LR_ABL & 8)) {
        if (lsh == 4) solve_packed(std::integral_constant<int, 16>{});
        else if (lsh == 5) solve_packed(std::integral_constant<int, 32>{});
        else if (rfl(n_c) <= 48) solve(std::integral_constant<int, 48>{});
        else solve(std::integral_constant<int, NP>{});
      }
    }
    __syncthreads();
    // 8. q_r = g_r - V'_r^T (D^1/2 z)_r
    {
      const int t = tid & (KP - 1), hh = tid >> 7, sl = 1 << lsh;
      for (int r = hh; r < (64 >> lsh); r += 2) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        const int j0 = r << lsh, j1 = min(j0 + sl, 32 * nrt);
        for (int j = j0; j < j1; j += 8) {
          const float4 ha = *reinterpret_cast<const float4*>(sH + j), hb = *reinterpret_cast<const float4*>(sH + j + 4);
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; e++) v[e] = sV[(j + e) * LD + t];
          s0 = fmaf(ha.x, v[0], s0); s1 = fmaf(ha.y, v[1], s1); s2 = fmaf(ha.z, v[2], s2); s3 = fmaf(ha.w, v[3], s3);
          s0 = fmaf(hb.x, v[4], s0); s1 = fmaf(hb.y, v[5], s1); s2 = fmaf(hb.z, v[6], s2); s3 = fmaf(hb.w, v[7], s3);
        }
        sQv[r * KP + t] = sGv[r * KP + t] - ((s0 + s1) + (s2 + s3));
      }
    }
    __syncthreads();
    // 9. y_r = M q_r = sum_j Mt[j][.] q_r[j]  (rows of Mt are coalesced and read once for all rows of the pass; the two halves of
    // the workgroup split j)
    {
      const int i2 = tid & (KP - 1), hj = tid >> 7;
      const float* Mtp = Mt + (size_t)(64 * hj) * KP + i2;
      asm volatile("" : "+v"(Mtp));   // (opaque: these loads must not be hoisted out of the row loop either)
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
      for (int j0 = 0; j0 < ((RSP_LR_ABL & 16) ? 0 : 64); j0 += 32) {   // 32 independent L2 reads in flight per thread
        float m[32];
#pragma unroll
        for (int e = 0; e < 32; e++) m[e] = Mtp[(size_t)(j0 + e) * KP];
        const float* qv = sQv + 64 * hj + j0;
        if (lsh == 6) {
#pragma unroll
          for (int e = 0; e < 32; e++) s0 = fmaf(m[e], qv[e], s0);
        } else if (lsh == 5) {
#pragma unroll
          for (int e = 0; e < 32; e++) {
            s0 = fmaf(m[e], qv[e], s0);
            s1 = fmaf(m[e], qv[KP + e], s1);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 32; e++) {
            s0 = fmaf(m[e], qv[e], s0);
            s1 = fmaf(m[e], qv[KP + e], s1);
            s2 = fmaf(m[e], qv[2 * KP + e], s2);
            s3 = fmaf(m[e], qv[3 * KP + e], s3);
          }
        }
      }
      float* dst = (hj ? sP : sY) + i2;
      dst[0] = s0; dst[KP] = s1; dst[2 * KP] = s2; dst[3 * KP] = s3;
    }
    __syncthreads();
    float lt = 0.f, yy = 0.f;
    if (tid < KP) {
      for (int r = 0; r < (64 >> lsh); r++) {
        const int rid = __builtin_amdgcn_readlane(rid_c, r << lsh);
        if (rid >= 0) {
          const float y = sY[r * KP + tid] + sP[r * KP + tid];
          if (tid < k) {
            a.Y[(size_t)rid * k + tid] = y;
            yy = fmaf(y, y, yy);
          }
        }
      }
    }
    // 10. loss: x_j . y = v_j . q
    {
      const int j = tid >> 2, part = tid & 3;
      float s = 0.f;
      if (j < 32 * nrt) {
        const float* vr = sV + j * LD + 32 * part;
        const float* qr = sQv + (j >> lsh) * KP + 32 * part;
#pragma unroll 8
        for (int e = 0; e < 32; e++) s = fmaf(vr[e], qr[e], s);
      }
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      if (part == 0 && j < 32 * nrt) {
        const float dlt = 1.f - s;
        lt = sC[j] * dlt * dlt;   // (empty slots: c = 0)
      }
    }
    const float lsum = wave_sum(lt), ysum = wave_sum(yy);
    if (lane == 0) wloss += (double)lsum + a.lambda_loss * (double)ysum;
    // shift the look-ahead
    rid_c = rid_n; n_c = n_n;
    rid_n = rid_nn; p1_n = p1_nn; n_n = n_nn;
    rid_nn = rid_n3;
    id_c = id_nx; c_c = c_nx;
  }
  __syncthreads();
  if (lane == 0) sRed[wv] = wloss;
  __syncthreads();
  if (tid == 0) a.loss_partials[loss_slot0 + blockIdx.x] = (sRed[0] + sRed[1]) + (sRed[2] + sRed[3]);
}


// ---- one WAVE per pass (round 4) ---------------------------------------------------------------------------------------
// The workgroup kernel above keeps four waves in step through a dozen barriers per pass and solves the n x n system on ONE of
// them (SQ counters on config 4: waves issuing 24 % of their cycles, waiting 69 %).  Here a pass -- the same 64 slots -- belongs
// to one wave from the gather to the loss, nothing is shared between waves but a read-only copy of the terms of M^T in LDS,
// and there is no barrier: eight passes are in flight per CU, each in a different phase.  The algebra is regrouped so that
// nothing has to be reduced over the lanes that hold the slots:
//     V'^T = M^T X_nnz^T            matrix cores; accumulator layout = lane: slot, registers: factor dimensions
//     T    = V' V'^T  (n x n)       matrix cores, both operands are the SAME registers (the fp16 terms of the tiles of V'^T)
//     S    = I + D^1/2 T D^1/2,  h = D^1/2 T c,  z = S^-1 h          lane i = slot i: T is symmetric, so the accumulator tiles
//                                   (lane: column, registers: rows) ARE rows spread over the lane pair (n, n + 32): one
//                                   v_permlane32_swap per register pair completes them -- T never touches LDS either
//     e    = c - D^1/2 z,   x_j . y = (T e)_j = (T c)_j - (h_j - z_j) / sqrt(c_j - 1)  (loss),   y = M V'^T e = P^T e
//     P    = V' M^T                 matrix cores again (A = the terms of V'^T, B = MT16); accumulator layout = lane: factor
//                                   dimension, registers: slots -> y is a sum over REGISTERS with coefficients e, no reduction
// g, h, q of the formulation above never exist as k-vectors.  A lane (n, hf) = (lane % 32, lane / 32) gathers ITS slots'
// vectors straight into the B-operand layout (slot n and slot 32 + n, the factor dimensions 16 ch + 8 hf ... + 7 of chunk
// ch: two 16-byte loads per slot and chunk), so X_nnz never touches LDS.
struct LrwSmem {
  static constexpr int LH = kLrLh;                              // halves per row of the copies of M16 and MT16
  static constexpr size_t m16_bytes = (size_t)2 * 128 * LH * 2;
  static constexpr int wave_floats = 64;                        // e
  static constexpr size_t bytes = 2 * m16_bytes + (size_t)8 * wave_floats * 4;
};

// r *= u(lane E of this lane's row of 16 lanes)
template <int E>
__device__ __forceinline__ void mul_row_bcast(float& r, const float u) {
  asm("v_mul_f32_dpp %0, %1, %0 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(u), "n"(E));
}
// copies of the value held one per lane such that lane t of a GROUP of SL lanes is lane t % 16 of rep[t / 16] in every lane
// of the group (SL = 64: the wave; 32; 16: a row of 16 lanes)
template <int SL>
__device__ __forceinline__ void lrw_group(const float v, float (&rep)[4]) {
  if constexpr (SL == 64) {
    rows_to_all<4>(v, rep);
  } else if constexpr (SL == 32) {
    const unsigned uu = __float_as_uint(v);
    const auto sw = __builtin_amdgcn_permlane16_swap(uu, uu, false, false);   // rows (0, 0, 2, 2) and (1, 1, 3, 3)
    rep[0] = __uint_as_float(sw[0]);
    rep[1] = __uint_as_float(sw[1]);
    rep[2] = rep[3] = 0.f;
  } else {
    rep[0] = v;
    rep[1] = rep[2] = rep[3] = 0.f;
  }
}
// 16 bytes at p + OFF, not waited for (the caller counts)
template <int OFF>
__device__ __forceinline__ void lrw_ld16(f32x4& d, const float* p) {
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(d) : "v"(p), "n"(OFF));
}
__device__ __forceinline__ void lrw_wait_all(f32x4 (&x)[8][2]) {   // (tied to the registers: nothing reads them before the wait)
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1]), "+v"(x[2][0]), "+v"(x[2][1]), "+v"(x[3][0]), "+v"(x[3][1]),
                 "+v"(x[4][0]), "+v"(x[4][1]), "+v"(x[5][0]), "+v"(x[5][1]), "+v"(x[6][0]), "+v"(x[6][1]), "+v"(x[7][0]), "+v"(x[7][1])
               :: "memory");
}
__device__ __forceinline__ f16x8 lrw_pack(const unsigned a, const unsigned b, const unsigned c, const unsigned d) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = {a, b, c, d};
  return __builtin_bit_cast(f16x8, v);
}

// (S + padd I) z = u in registers, S symmetric, lane i = row i of its system: rl[t] = S[i][base + t] (diagonal entries are only
// ever read as pivots, which is where padd is added: the identity of S = I + D^1/2 T D^1/2, or lambda).  SL = 64: one system on
// the whole wave, NS <= 64 columns held (the register solve of the workgroup kernel: the pivot column's rows of 16 lanes are
// copied into every row once per pivot, the multipliers are DPP row broadcasts inside the FMAs, the next pivot column is served
// first and by v_readlane).  SL = 32 / 16: 64 / SL independent systems, one per group of SL lanes, register t = column t of the
// group, every broadcast local to the group.  A `unit` lane is not a row of the system: it holds a row of right-hand-side kind,
// every pivot eliminates it, its column is masked out of the broadcasts, and what the forward pass leaves in its u is returned
// in u_fwd (its z is 0).  rl is destroyed.
template <int SL, int NS>
__device__ __forceinline__ void lrw_solve(float (&rl)[NS], float u, const int i, const bool unit, const float padd, float& z,
                                          float& u_fwd) {
  const int il = i & (SL - 1);
  const int ie = unit ? 1024 : il;   // "row index" for the elimination: a unit lane is below every pivot
  float dinv = 1.f;
  z = 0.f;
  if constexpr (SL == 64) {
    float pj = readlane_f(rl[0], 0) + readlane_f(padd, 0);
    lr_sfor<NS>([&](auto jt) {
      constexpr int j = decltype(jt)::value;
      const float inv = __builtin_amdgcn_rcpf(pj);
      const float uj = readlane_f(u, j);
      if (i == j) dinv = inv;
      const float lij = ie > j ? rl[j] * inv : 0.f;
      u = fmaf(-lij, uj, u);
      if constexpr (j + 1 < NS) {
        float cj = unit ? 0.f : rl[j];   // column j as the symmetric system has it
        rl[j + 1] = fmaf(-lij, readlane_f(cj, j + 1), rl[j + 1]);
        pj = readlane_f(rl[j + 1], j + 1) + readlane_f(padd, j + 1);
        if constexpr (j + 2 < NS) {
          float rep[4];
          dpp_ready(cj);
          rows_to_all<4>(cj, rep);
          dpp_ready(rep[0], rep[1], rep[2], rep[3]);
          lr_sfor<NS - j - 2>([&](auto ct) {
            constexpr int c = j + 2 + decltype(ct)::value;
            fnma_row_bcast<c % 16>(rl[c], rep[c / 16], lij);
          });
        }
      }
    });
    u_fwd = u;
    if (unit) dinv = 0.f;
    float bacc = 0.f;
#pragma unroll
    for (int c = NS - 1; c >= 0; c--) {
      if (i == c) z = (u - bacc) * dinv;
      const float zc = readlane_f(z, c);
      bacc = fmaf(i < c ? rl[c] : 0.f, zc, bacc);
    }
  } else {
    static_assert(NS == SL || SL == 64, "packed systems hold all their columns");
    lr_sfor<SL>([&](auto tt) {
      constexpr int t = decltype(tt)::value;
      float rep[4], ur[4];
      float ct = unit ? 0.f : rl[t];   // column t as the symmetric system has it
      dpp_ready(ct, u);
      lrw_group<SL>(ct, rep);
      lrw_group<SL>(u, ur);
      dpp_ready(rep[0], rep[1], ur[0], ur[1]);
      const float pv = lr_row_bcast<t % 16>(rep[t / 16]) + padd;   // (padd is uniform inside a group)
      const float inv = __builtin_amdgcn_rcpf(pv);
      if (il == t) dinv = inv;
      const float lij = ie > t ? rl[t] * inv : 0.f;
      fnma_row_bcast<t % 16>(u, ur[t / 16], lij);
      lr_sfor<SL - t - 1>([&](auto ct2) {
        constexpr int c = t + 1 + decltype(ct2)::value;
        fnma_row_bcast<c % 16>(rl[c], rep[c / 16], lij);
      });
    });
    u_fwd = u;
    if (unit) dinv = 0.f;
    float bacc = 0.f;
    lr_sfor<SL>([&](auto tt) {
      constexpr int c = SL - 1 - decltype(tt)::value;
      if (il == c) z = (u - bacc) * dinv;
      float zr[4];
      lrw_group<SL>(z, zr);
      dpp_ready(zr[0], zr[1]);
      const float m = il < c ? -rl[c] : 0.f;
      fnma_row_bcast<c % 16>(bacc, zr[c / 16], m);
    });
  }
}

#ifdef RSP_LRW_PROF
// dev builds: s_memtime ticks per phase, summed over the waves of a class (read by rsparse_hip_dev_lrw_prof)
__device__ unsigned long long g_lrw_prof[4][8];
#define LRW_TICK(k, dep)                                  \
  {                                                       \
    __builtin_amdgcn_sched_barrier(0);                    \
    asm volatile("s_nop 0" ::"v"(dep) : "memory");        \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    __builtin_amdgcn_sched_barrier(0);                    \
    prof_t[k] += now_ - prof_last;                        \
    prof_last = now_;                                     \
  }
#else
#define LRW_TICK(k, dep)
#endif

// One instantiation per class of passes -- SL = slots per row (64, 32, 16), NS = columns of the system held (48 for the rows
// of 33..48 non-zeros: the list is longest first, so they are a range of passes too) -- and one launch per class over its
// passes [pass_lo, pass_hi): every launch's code fits the instruction cache (all four unrolled solves in one kernel were
// 100 KB of code for eight waves in different phases) and has the register allocation of its own solve.
template <int KP, int SL, int NS>
__global__ __launch_bounds__(512) void als_chol_lrw_kernel(AlsArgs a, const int32_t* __restrict__ rows, int n_rows, int n64,
                                                           int n32, int pass_lo, int pass_hi,
                                                           const _Float16* __restrict__ M16,
                                                           const _Float16* __restrict__ MT16,
                                                           const unsigned* __restrict__ flags, int loss_slot0) {
  static_assert(KP == 128, "written for rank 128");
  constexpr int lsh = SL == 64 ? 6 : (SL == 32 ? 5 : 4);
  constexpr int LH = LrwSmem::LH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  _Float16* sM = reinterpret_cast<_Float16*>(smem);   // [2][KP][LH]: the two fp16 terms of Mt * 2^eM (A operand of V'^T)
  __shared__ double sRed[8];
  const int tid = threadIdx.x, lane = tid & 63, wv = rfl(tid >> 6);
  _Float16* sMT = sM + (size_t)2 * KP * LH;           // [2][KP][LH]: MT16 (B operand of P = V' M^T)
  float* sE = reinterpret_cast<float*>(smem + 2 * LrwSmem::m16_bytes) + wv * LrwSmem::wave_floats;
  if (flags[0] != 0) return;   // some confidence < 1 or XtX not positive definite: wrmf_chol.hip takes these rows
  for (int e = tid; e < 2 * KP * 16; e += 512) {   // 16-byte pieces
    const int t = e >> 11, r = (e >> 4) & (KP - 1), p = e & 15;
    *reinterpret_cast<uint4*>(sM + (size_t)(t * KP + r) * LH + 8 * p) =
        *reinterpret_cast<const uint4*>(M16 + (size_t)t * KP * KP + (size_t)r * KP + 8 * p);
    *reinterpret_cast<uint4*>(sMT + (size_t)(t * KP + r) * LH + 8 * p) =
        *reinterpret_cast<const uint4*>(MT16 + (size_t)t * KP * KP + (size_t)r * KP + 8 * p);
  }
  __syncthreads();
  const int ex = lr_scale_exp(fmaxf(__uint_as_float(flags[-2]), 1e-30f)), eM = (int)flags[1];
  const float sx = lr_pow2(ex), cs = lr_pow2(254 - ex) * lr_pow2(254 - eM), cm = lr_pow2(254 - eM);

  const int P64 = n64, P32 = (n32 + 1) >> 1;
  struct PassGeo { int lsh, li, lim; };
  auto geo = [&](const int pp) {   // (all passes of a launch are of its class)
    PassGeo g;
    g.lsh = lsh;
    if constexpr (SL == 64) { g.li = pp; g.lim = n64; }
    else if constexpr (SL == 32) { g.li = n64 + 2 * (pp - P64) + (lane >> 5); g.lim = n64 + n32; }
    else { g.li = n64 + n32 + 4 * (pp - P64 - P32) + (lane >> 4); g.lim = n_rows; }
    if (pp >= pass_hi) g.lim = 0;
    return g;
  };

  double wloss = 0.0;
  // row metadata ahead of the passes, as in the workgroup kernel (all per lane: lane j works for the row of slot j)
  const int G = gridDim.x * 8;
  int it = pass_lo + blockIdx.x * 8 + wv;
  int rid_c = -1, n_c = 0, rid_n = -1, p1_n = 0, n_n = 0, rid_nn = -1;
  int id_c = 0;
  float c_c = 1.f;
  {
    const PassGeo g0 = geo(it), g1 = geo(it + G), g2 = geo(it + 2 * G);
    int p1_c = 0;
    if (g0.li < g0.lim) {
      rid_c = rows[g0.li];
      p1_c = a.col_ptrs[rid_c];
      n_c = a.col_ptrs[rid_c + 1] - p1_c;
    }
    if (g1.li < g1.lim) {
      rid_n = rows[g1.li];
      p1_n = a.col_ptrs[rid_n];
      n_n = a.col_ptrs[rid_n + 1] - p1_n;
    }
    if (g2.li < g2.lim) rid_nn = rows[g2.li];
    const int nz = lane & ((1 << g0.lsh) - 1);
    if (nz < n_c) {
      id_c = a.row_idx[p1_c + nz];
      c_c = a.vals[p1_c + nz];
    }
  }
#ifdef RSP_LRW_PROF
  unsigned long long prof_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long prof_last = __builtin_readcyclecounter();
#endif
  for (; it < pass_hi; it += G) {
    int ln = lane;
    asm volatile("" : "+v"(ln));   // the lane id as this pass sees it (keeps lane-dependent addresses inside the pass)
    const int nz_c = ln & ((1 << lsh) - 1);
    const bool valid = nz_c < n_c;
    const unsigned long long vmask = __ballot(valid);
    int id_nx = 0, p1_nn = 0, n_nn = 0, rid_n3 = -1;
    float c_nx = 1.f;
    {
      const PassGeo g1 = geo(it + G), g3 = geo(it + 3 * G);
      const int nz1 = ln & ((1 << g1.lsh) - 1);
      if (nz1 < n_n) {
        id_nx = a.row_idx[p1_n + nz1];
        c_nx = a.vals[p1_n + nz1];
      }
      if (rid_nn >= 0) {
        p1_nn = a.col_ptrs[rid_nn];
        n_nn = a.col_ptrs[rid_nn + 1] - p1_nn;
      }
      if (g3.li < g3.lim) rid_n3 = rows[g3.li];
    }
    const float cval = valid ? c_c : 0.f;
    const float sq = valid ? sqrtf(fmaxf(c_c - 1.f, 0.f)) : 0.f;
    const int n = ln & 31, hf = ln >> 5;

    // ---- 1. gather + V'^T = Mt X_nnz^T: acc[ob][st] = dims 32 ob.. x slots 32 st.. ----
    // Every load of the pass is issued at once (32 per lane); the chunks are then taken in DESCENDING order: Mt is lower
    // triangular, so chunk ch only feeds the block rows ob >= ch / 2 -- the accumulators come to life (32 registers per block
    // row) as the gathered registers are consumed, and the two never exceed 160 registers together.
    f32x16 acc[4][2];
    {
      // (a slot beyond its row has index 0 and scale 0: its loads hit row 0 of X, its vector is zero)
      const auto idsw = __builtin_amdgcn_permlane32_swap((unsigned)id_c, (unsigned)id_c, false, false);
      const float sx0 = ((vmask >> n) & 1) ? sx : 0.f, sx1 = ((vmask >> (32 + n)) & 1) ? sx : 0.f;
      const float* x0 = a.X + (size_t)idsw[0] * KP + 8 * hf;
      const float* x1 = a.X + (size_t)idsw[1] * KP + 8 * hf;
      // (from inline asm: hipcc sinks plain loads to their uses -- six in flight, a round trip per chunk -- whatever
      //  scheduling barrier follows them; the one wait below covers everything this wave has in flight)
      f32x4 xr[2][8][2];   // [slot tile][chunk][two 16-byte pieces]
      lr_sfor<8>([&](auto ct) {
        constexpr int c8 = 7 - decltype(ct)::value;
        lrw_ld16<64 * c8>(xr[0][c8][0], x0);
        lrw_ld16<64 * c8 + 16>(xr[0][c8][1], x0);
        lrw_ld16<64 * c8>(xr[1][c8][0], x1);
        lrw_ld16<64 * c8 + 16>(xr[1][c8][1], x1);
      });
      LRW_TICK(0, ln);
      lrw_wait_all(xr[0]);
      lrw_wait_all(xr[1]);
      LRW_TICK(1, xr[0][0][0].x);
      const _Float16* ap0 = sM + (size_t)n * LH + 8 * hf;
      lr_sfor<8>([&](auto cht) {
        constexpr int ch = 7 - decltype(cht)::value;
        if constexpr (ch % 2 == 1) {   // first use of block row ch / 2
#pragma unroll
          for (int e = 0; e < 16; e++) acc[ch / 2][0][e] = acc[ch / 2][1][e] = 0.f;
        }
        f16x8 bh[2], bl[2];
#pragma unroll
        for (int st = 0; st < 2; st++) {
          const f32x4 p0 = xr[st][ch][0], p1 = xr[st][ch][1];
          const float sxs = st ? sx1 : sx0;
          unsigned h0, h1, h2, h3, l0, l1, l2, l3;
          lr_split(p0.x * sxs, p0.y * sxs, h0, l0);
          lr_split(p0.z * sxs, p0.w * sxs, h1, l1);
          lr_split(p1.x * sxs, p1.y * sxs, h2, l2);
          lr_split(p1.z * sxs, p1.w * sxs, h3, l3);
          bh[st] = lrw_pack(h0, h1, h2, h3);
          bl[st] = lrw_pack(l0, l1, l2, l3);
        }
        lr_sfor<4 - ch / 2>([&](auto obt) {
          constexpr int ob = ch / 2 + decltype(obt)::value;   // Mt is lower triangular: block row ob ends at chunk 2 ob + 1
          const _Float16* ap = ap0 + (size_t)(32 * ob) * LH + 16 * ch;
          const f16x8 ah = *reinterpret_cast<const f16x8*>(ap);
          const f16x8 al = *reinterpret_cast<const f16x8*>(ap + (size_t)KP * LH);
          if constexpr (!(RSP_LRW_ABL & 1)) {
            acc[ob][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[0], acc[ob][0], 0, 0, 0);
            acc[ob][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[1], acc[ob][1], 0, 0, 0);
            acc[ob][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[0], acc[ob][0], 0, 0, 0);
            acc[ob][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[1], acc[ob][1], 0, 0, 0);
            acc[ob][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[0], acc[ob][0], 0, 0, 0);
            acc[ob][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[1], acc[ob][1], 0, 0, 0);
          } else {   // (timing only)
            acc[ob][0][0] += (float)ah[0] * (float)bh[0][0] + (float)al[1] * (float)bl[0][1];
            acc[ob][1][0] += (float)ah[0] * (float)bh[1][0] + (float)al[1] * (float)bl[1][1];
          }
        });
      });
    }
    LRW_TICK(2, acc[0][0][0] + acc[3][1][0]);
    // ---- 2. the fp16 terms of V'^T * 2^ew (th / tl[ob][st][g]: the 16 factor dimensions of chunk 2 ob + g, in the order the
    // accumulator holds them) ----
    float wm = 0.f;
#pragma unroll
    for (int ob = 0; ob < 4; ob++)
#pragma unroll
      for (int st = 0; st < 2; st++)
#pragma unroll
        for (int e = 0; e < 16; e++) wm = fmaxf(wm, fabsf(acc[ob][st][e]));
    for (int o = 32; o > 0; o >>= 1) wm = fmaxf(wm, __shfl_xor(wm, o));
    const int ew = lr_scale_exp(fmaxf(wm, 1e-30f));
    const float fw = lr_pow2(ew), c2 = cs * lr_pow2(254 - ew);   // terms = V' / c2
    f16x8 th[4][2][2], tl[4][2][2];
#pragma unroll
    for (int ob = 0; ob < 4; ob++)
#pragma unroll
      for (int st = 0; st < 2; st++)
#pragma unroll
        for (int g = 0; g < 2; g++) {
          unsigned hh[4], ll[4];
#pragma unroll
          for (int p = 0; p < 4; p++)
            lr_split(acc[ob][st][8 * g + 2 * p] * fw, acc[ob][st][8 * g + 2 * p + 1] * fw, hh[p], ll[p]);
          th[ob][st][g] = lrw_pack(hh[0], hh[1], hh[2], hh[3]);
          tl[ob][st][g] = lrw_pack(ll[0], ll[1], ll[2], ll[3]);
        }
    // ---- 3. T = V' V'^T / c2^2, all four tiles (packed rows: the two diagonal ones, the others only feed lanes that do not read them); tile (a, b) at lane (n, hf), register v = T[32 a + rho(v, hf)][32 b + n] with
    // rho(v, hf) = 8 (v / 4) + 4 hf + v % 4 -- by symmetry ROW 32 b + n at the columns 32 a + rho(v, hf).  Lane (n, 0) is slot n
    // and keeps its tiles (a, 0), lane (n, 1) is slot 32 + n and keeps its tiles (a, 1); the other two go to the partner lane:
    // afterwards ra[a][v] = column 32 a + rho(v, 0) and rb[a][v] = column 32 a + rho(v, 1) of the lane's own row, in every lane ----
    float rl[NS];   // rl[t] = T[slot][base + t] / c2^2
    {
      f32x16 t[2][2];
#pragma unroll
      for (int e = 0; e < 16; e++) t[0][0][e] = t[0][1][e] = t[1][0][e] = t[1][1][e] = 0.f;
#pragma unroll
      for (int ob = 0; ob < 4; ob++)
#pragma unroll
        for (int g = 0; g < 2; g++) {
          const f16x8 a0h = th[ob][0][g], a0l = tl[ob][0][g], a1h = th[ob][1][g], a1l = tl[ob][1][g];
          if constexpr (!(RSP_LRW_ABL & 2)) {
            t[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, a0h, t[0][0], 0, 0, 0);
            if constexpr (SL == 64) t[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, a0h, t[1][0], 0, 0, 0);
            if constexpr (SL == 64) t[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, a1h, t[0][1], 0, 0, 0);
            t[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, a1h, t[1][1], 0, 0, 0);
            t[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, a0l, t[0][0], 0, 0, 0);
            if constexpr (SL == 64) t[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, a0l, t[1][0], 0, 0, 0);
            if constexpr (SL == 64) t[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, a1l, t[0][1], 0, 0, 0);
            t[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, a1l, t[1][1], 0, 0, 0);
            t[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, a0h, t[0][0], 0, 0, 0);
            if constexpr (SL == 64) t[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, a0h, t[1][0], 0, 0, 0);
            if constexpr (SL == 64) t[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, a1h, t[0][1], 0, 0, 0);
            t[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, a1h, t[1][1], 0, 0, 0);
          } else {
            t[0][0][0] += (float)a0h[0]; t[1][0][0] += (float)a0l[0]; t[0][1][0] += (float)a1h[0]; t[1][1][0] += (float)a1l[0];
          }
        }
      // (the columns of the lane's own group only: rl[t] = T[slot][base + t], base = slot - slot % SL)
      auto swp = [&](const int ta, const int v, float& c_lo, float& c_hi) {   // columns 32 ta + rho(v, 0) and + rho(v, 1)
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(t[ta][0][v]), __float_as_uint(t[ta][1][v]), false, false);
        c_lo = __uint_as_float(sw[0]);
        c_hi = __uint_as_float(sw[1]);
      };
      if constexpr (SL == 64) {
#pragma unroll
        for (int ta = 0; ta < 2; ta++)
#pragma unroll
          for (int v = 0; v < 16; v++) {
            const int c0 = 32 * ta + 8 * (v >> 2) + (v & 3);
            float x0, x1;
            swp(ta, v, x0, x1);
            if (c0 < NS) rl[c0] = x0;
            if (c0 + 4 < NS) rl[c0 + 4] = x1;
          }
      } else if constexpr (SL == 32) {
#pragma unroll
        for (int v = 0; v < 16; v++) {   // slots 0..31: the tiles ta = 0, slots 32..63: ta = 1
          const int c0 = 8 * (v >> 2) + (v & 3);
          float a0, a1, b0, b1;
          swp(0, v, a0, a1);
          swp(1, v, b0, b1);
          rl[c0] = hf ? b0 : a0;
          rl[c0 + 4] = hf ? b1 : a1;
        }
      } else {
        const int gq = (ln >> 4) & 1;
#pragma unroll
        for (int v = 0; v < 8; v++) {   // group (2 ta + gq): the rows v and v + 8 of the tiles ta
          const int c0 = 8 * (v >> 2) + (v & 3);
          float a0, a1, b0, b1, c0v, c1v, d0, d1;
          swp(0, v, a0, a1);
          swp(0, v + 8, b0, b1);
          swp(1, v, c0v, c1v);
          swp(1, v + 8, d0, d1);
          const float lo0 = gq ? b0 : a0, lo1 = gq ? b1 : a1, hi0 = gq ? d0 : c0v, hi1 = gq ? d1 : c1v;
          rl[c0] = hf ? hi0 : lo0;
          rl[c0 + 4] = hf ? hi1 : lo1;
        }
      }
    }
    LRW_TICK(3, rl[0] + rl[NS - 1]);
    // ---- 4. lane i = slot i: S z = h in registers, e, the loss terms ----
    const float sqs = sq * c2;   // (D^1/2 T D^1/2)_ic = r[c] sqs_i sqs_c
    float e_i = 0.f, p_i = 0.f;   // p = x_j . y
    {
      const int i = ln, il = i & (SL - 1);
      // A slot of confidence exactly 1 (d = 0) has no row in W: row and column of S are those of the identity.  Its lane keeps
      // its row of T instead, scaled like a right-hand side (T_ic sqrt(d_c)), and lets EVERY pivot eliminate it: what the
      // forward pass leaves in its u is (T c)_i - T_i D^1/2 S^-1 h = (T e)_i = x_i . y, the slot's loss term, for free.  (The
      // bench matrix has half of its confidences at 1.)  The broadcast copies of a pivot column are masked at such lanes, so
      // the other rows never see them: the system stays the symmetric one.
      const bool unit = valid && !(sq > 0.f);
      float repc[4], repq[4];
      lrw_group<SL>(cval, repc);
      lrw_group<SL>(sqs, repq);
      dpp_ready(repc[0], repc[1], repc[2], repc[3]);
      dpp_ready(repq[0], repq[1], repq[2], repq[3]);
      const float rowf = unit ? c2 : sqs;
      float tc = 0.f;   // -(T c)_i / c2^2
      lr_sfor<NS>([&](auto tt) {
        constexpr int t = decltype(tt)::value;
        fnma_row_bcast<t % 16>(tc, repc[t / 16], rl[t]);
        rl[t] *= rowf;
        mul_row_bcast<t % 16>(rl[t], repq[t / 16]);   // (D^1/2 T D^1/2)_{i, base + t}; the identity is added at the pivots
      });
      tc = -(tc * c2) * c2;   // (T c)_i
      float u = unit ? tc : tc * sq;   // h_i
      float z = 0.f;
      if constexpr (RSP_LRW_ABL & 4) {
        z = u + rl[0] + rl[NS - 1];
      } else {
        lrw_solve<SL, NS>(rl, u, i, unit, 1.f, z, p_i);
      }
      e_i = cval - sq * z;   // (slots beyond the row: c = 0, sq = 0; unit slots: z = 0)
      // x_i . y = (T e)_i, and D^1/2 T e = D^1/2 T c - D^1/2 T D^1/2 z = h - (S - I) z = z
      if (!unit) p_i = z * __builtin_amdgcn_rcpf(fmaxf(sq, 1e-30f));
    }
    // (the look-ahead requested at the top has long arrived: waited for HERE, before this pass's stores are in the queue --
    //  hipcc waits for loop-carried loads with vmcnt(0) at their first use, which at the top of the next pass would be a wait for
    //  those stores)
    asm volatile("" : "+v"(id_nx), "+v"(c_nx), "+v"(p1_nn), "+v"(n_nn), "+v"(rid_n3));
    LRW_TICK(4, e_i + p_i);
    wave_sync();   // (the previous pass has read its e)
    sE[ln] = e_i;
    wave_sync();
    // ---- 5. y = P^T e, P = V' M^T one block of 32 columns at a time ----
    float yy = 0.f;
    {
      float4 ec[2][4];
#pragma unroll
      for (int st = 0; st < 2; st++)
#pragma unroll
        for (int q = 0; q < 4; q++) ec[st][q] = *reinterpret_cast<const float4*>(sE + 32 * st + 8 * q + 4 * hf);
      const _Float16* bp0 = sMT + (size_t)n * LH + 8 * hf;
      const float pscale = c2;
      lr_sfor<4>([&](auto cbt) {
        constexpr int cb = decltype(cbt)::value;
        f32x16 p0, p1;
#pragma unroll
        for (int e = 0; e < 16; e++) p0[e] = p1[e] = 0.f;
        lr_sfor<8 - 2 * cb>([&](auto cht) {
          constexpr int chunk = 2 * cb + decltype(cht)::value;   // Mt[kk][c] = 0 for kk < c
          constexpr int ob = chunk / 2, g = chunk % 2;
          const _Float16* bp = bp0 + (size_t)(32 * cb) * LH + 16 * chunk;
          const f16x8 bh = *reinterpret_cast<const f16x8*>(bp);
          const f16x8 bl = *reinterpret_cast<const f16x8*>(bp + (size_t)KP * LH);
          if constexpr (!(RSP_LRW_ABL & 8)) {
            p0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(th[ob][0][g], bh, p0, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(th[ob][1][g], bh, p1, 0, 0, 0);
            p0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(th[ob][0][g], bl, p0, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(th[ob][1][g], bl, p1, 0, 0, 0);
            p0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(tl[ob][0][g], bh, p0, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(tl[ob][1][g], bh, p1, 0, 0, 0);
          } else {
            p0[0] += (float)th[ob][0][g][0] * (float)bh[0] + (float)tl[ob][0][g][0] * (float)bl[0];
            p1[0] += (float)th[ob][1][g][0] * (float)bh[1] + (float)tl[ob][1][g][0] * (float)bl[1];
          }
        });
        // sums over the slots of this lane half, by 16-slot group: q[2 st + (a >> 1)], a = 8-row block of the tile
        float q[4];
#pragma unroll
        for (int a2 = 0; a2 < 4; a2++) {
          const float4 e0 = ec[0][a2], e1 = ec[1][a2];
          const float s0 = fmaf(e0.x, p0[4 * a2], fmaf(e0.y, p0[4 * a2 + 1], fmaf(e0.z, p0[4 * a2 + 2], e0.w * p0[4 * a2 + 3])));
          const float s1 = fmaf(e1.x, p1[4 * a2], fmaf(e1.y, p1[4 * a2 + 1], fmaf(e1.z, p1[4 * a2 + 2], e1.w * p1[4 * a2 + 3])));
          if (a2 & 1) { q[a2 >> 1] += s0; q[2 + (a2 >> 1)] += s1; }
          else { q[a2 >> 1] = s0; q[2 + (a2 >> 1)] = s1; }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const unsigned uq = __float_as_uint(q[r]);
          const auto sw = __builtin_amdgcn_permlane32_swap(uq, uq, false, false);
          q[r] = ((__uint_as_float(sw[0]) + __uint_as_float(sw[1])) * pscale) * cm;
        }
        // rows of the pass: one (all four groups), two (groups 0 + 1, 2 + 3) or four
        float yr[4];
        if constexpr (SL == 64) { yr[0] = (q[0] + q[1]) + (q[2] + q[3]); yr[1] = yr[2] = yr[3] = 0.f; }
        else if constexpr (SL == 32) { yr[0] = q[0] + q[1]; yr[1] = q[2] + q[3]; yr[2] = yr[3] = 0.f; }
        else { yr[0] = q[0]; yr[1] = q[1]; yr[2] = q[2]; yr[3] = q[3]; }
#pragma unroll
        for (int r = 0; r < 4; r++) {
          if (r < (64 >> lsh)) {   // wave-uniform
            const int rid = __builtin_amdgcn_readlane(rid_c, r << lsh);
            if (rid >= 0 && hf == 0) {
              a.Y[(size_t)rid * KP + 32 * cb + n] = yr[r];
              yy = fmaf(yr[r], yr[r], yy);
            }
          }
        }
      });
    }
    LRW_TICK(5, yy);
    const float dlt = 1.f - p_i;
    const float loss_i = cval * dlt * dlt;
    const float lsum = wave_sum(loss_i), ysum = wave_sum(yy);
    wloss += (double)lsum + a.lambda_loss * (double)ysum;
    rid_c = rid_n; n_c = n_n;
    rid_n = rid_nn; p1_n = p1_nn; n_n = n_nn;
    rid_nn = rid_n3;
    id_c = id_nx; c_c = c_nx;
    LRW_TICK(6, (float)(id_c + rid_c) + c_c);
#ifdef RSP_LRW_PROF
    prof_t[7] += 1;
#endif
  }
#ifdef RSP_LRW_PROF
  if (lane == 0) {
    constexpr int cls = SL == 64 ? (NS == 64 ? 0 : 1) : (SL == 32 ? 2 : 3);
    for (int q = 0; q < 8; q++) atomicAdd(&g_lrw_prof[cls][q], prof_t[q]);
  }
#endif
  if (lane == 0) sRed[wv] = wloss;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    for (int w = 0; w < 8; w++) s += sRed[w];
    a.loss_partials[loss_slot0 + blockIdx.x] = s;
  }
}


// ---- explicit feedback, one wave per pass (round 4) ---------------------------------------------------------------------
// als_explicit<T>'s exact branch (inst/include/wrmf_explicit.hpp:103-108): lhs = X_nnz X_nnz^T + lambda_use I, rhs = X_nnz r.  For a
// row of n <= 64 ratings the k x k factorisation is wasted on a rank-n matrix plus a multiple of the identity; push the inverse
// through (no Gramian, so no M either):
//     y = X_nnz (lambda I_n + T)^-1 r,   T = X_nnz^T X_nnz (n x n),     r_j - x_j . y = lambda z_j  with z = (lambda I + T)^-1 r
// -- the loss terms are lambda z.  A pass is the 64 slots of the implicit kernel above (one row, two of <= 32 ratings, four of
// <= 16), a wave owns it from the gather to the loss: the gathered vectors (lane = slot) ARE both operands of T on the matrix
// cores, T reaches lane = row by lane swaps, (T + lambda I) z = r is lrw_solve, and y = sum_j z_j x_j is a sum over REGISTERS once
// the vectors have been turned to lane = coordinate -- by multiplying them with the identity on the matrix cores.  Ranks 64 and 128.
template <int NCH>
__device__ __forceinline__ void lrx_wait_all(f32x4 (&x)[NCH][2]) {
  if constexpr (NCH == 8) {
    lrw_wait_all(x);
  } else {
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1]), "+v"(x[2][0]), "+v"(x[2][1]), "+v"(x[3][0]), "+v"(x[3][1])
                 :: "memory");
  }
}

template <int KP, int SL, int NS>
__global__ __launch_bounds__(512) void als_chol_lrx_kernel(AlsArgs a, const int32_t* __restrict__ rows, int n_rows, int n64,
                                                           int n32, int pass_lo, int pass_hi,
                                                           const unsigned* __restrict__ stats, int loss_slot0) {
  static_assert(KP == 64 || KP == 128, "ranks 64 and 128");
  constexpr int lsh = SL == 64 ? 6 : (SL == 32 ? 5 : 4);
  constexpr int NCH = KP / 16, NB = KP / 32;
  __shared__ float sEall[8][64];
  __shared__ double sRed[8];
  const int tid = threadIdx.x, lane = tid & 63, wv = rfl(tid >> 6);
  float* sE = sEall[wv];
  const int ex = lr_scale_exp(fmaxf(__uint_as_float(stats[0]), 1e-30f));
  const float sx = lr_pow2(ex), ux = lr_pow2(254 - ex);   // scale of the fp16 terms and its inverse

  const int P64 = n64, P32 = (n32 + 1) >> 1;
  struct PassGeo { int lsh, li, lim; };
  auto geo = [&](const int pp) {   // (all passes of a launch are of its class)
    PassGeo g;
    g.lsh = lsh;
    if constexpr (SL == 64) { g.li = pp; g.lim = n64; }
    else if constexpr (SL == 32) { g.li = n64 + 2 * (pp - P64) + (lane >> 5); g.lim = n64 + n32; }
    else { g.li = n64 + n32 + 4 * (pp - P64 - P32) + (lane >> 4); g.lim = n_rows; }
    if (pp >= pass_hi) g.lim = 0;
    return g;
  };

  double wloss = 0.0;
  const int G = gridDim.x * 8;
  int it = pass_lo + blockIdx.x * 8 + wv;
  int rid_c = -1, n_c = 0, rid_n = -1, p1_n = 0, n_n = 0, rid_nn = -1;
  int id_c = 0;
  float c_c = 0.f;
  {
    const PassGeo g0 = geo(it), g1 = geo(it + G), g2 = geo(it + 2 * G);
    int p1_c = 0;
    if (g0.li < g0.lim) {
      rid_c = rows[g0.li];
      p1_c = a.col_ptrs[rid_c];
      n_c = a.col_ptrs[rid_c + 1] - p1_c;
    }
    if (g1.li < g1.lim) {
      rid_n = rows[g1.li];
      p1_n = a.col_ptrs[rid_n];
      n_n = a.col_ptrs[rid_n + 1] - p1_n;
    }
    if (g2.li < g2.lim) rid_nn = rows[g2.li];
    const int nz = lane & ((1 << g0.lsh) - 1);
    if (nz < n_c) {
      id_c = a.row_idx[p1_c + nz];
      c_c = a.vals[p1_c + nz];
    }
  }
  // the identity as B fragments: chunk cc of a block of 32 columns, element e of lane (n, hf) = (16 cc + 8 hf + e == n)
  f16x8 idf[2];
  {
    const int n0 = lane & 31, hf0 = lane >> 5;
#pragma unroll
    for (int cc = 0; cc < 2; cc++) {
      const bool mine = (n0 >> 4) == cc && ((n0 >> 3) & 1) == hf0;
      const int e1 = n0 & 7;   // 1.0 as fp16 = 0x3c00
      unsigned w[4];
#pragma unroll
      for (int q = 0; q < 4; q++) w[q] = (mine && (e1 >> 1) == q) ? ((e1 & 1) ? 0x3c000000u : 0x00003c00u) : 0u;
      idf[cc] = lrw_pack(w[0], w[1], w[2], w[3]);
    }
  }
  for (; it < pass_hi; it += G) {
    int ln = lane;
    asm volatile("" : "+v"(ln));   // the lane id as this pass sees it (keeps lane-dependent addresses inside the pass)
    const int nz_c = ln & ((1 << lsh) - 1);
    const bool valid = nz_c < n_c;
    const unsigned long long vmask = __ballot(valid);
    int id_nx = 0, p1_nn = 0, n_nn = 0, rid_n3 = -1;
    float c_nx = 0.f;
    {
      const PassGeo g1 = geo(it + G), g3 = geo(it + 3 * G);
      const int nz1 = ln & ((1 << g1.lsh) - 1);
      if (nz1 < n_n) {
        id_nx = a.row_idx[p1_n + nz1];
        c_nx = a.vals[p1_n + nz1];
      }
      if (rid_nn >= 0) {
        p1_nn = a.col_ptrs[rid_nn];
        n_nn = a.col_ptrs[rid_nn + 1] - p1_nn;
      }
      if (g3.li < g3.lim) rid_n3 = rows[g3.li];
    }
    const float rating = valid ? c_c : 0.f;
    // lambda_use of the lane's row (a group without a row: 1, its lanes are inert)
    const float lam_i = n_c > 0 ? (float)(a.lambda_loss * (a.dynamic_lambda ? (double)(float)n_c : 1.0)) : 1.f;
    const int n = ln & 31, hf = ln >> 5;

    // ---- 1. gather (lane (n, hf): slots n and 32 + n, the coordinates 16 ch + 8 hf .. + 7 of chunk ch) -> fp16 terms ----
    f16x8 th[NCH][2], tl[NCH][2];   // [chunk][slot tile]
    {
      const auto idsw = __builtin_amdgcn_permlane32_swap((unsigned)id_c, (unsigned)id_c, false, false);
      const float sx0 = ((vmask >> n) & 1) ? sx : 0.f, sx1 = ((vmask >> (32 + n)) & 1) ? sx : 0.f;
      const float* x0 = a.X + (size_t)idsw[0] * KP + 8 * hf;
      const float* x1 = a.X + (size_t)idsw[1] * KP + 8 * hf;
      f32x4 xr[2][NCH][2];
      lr_sfor<NCH>([&](auto ct) {
        constexpr int c8 = decltype(ct)::value;
        lrw_ld16<64 * c8>(xr[0][c8][0], x0);
        lrw_ld16<64 * c8 + 16>(xr[0][c8][1], x0);
        lrw_ld16<64 * c8>(xr[1][c8][0], x1);
        lrw_ld16<64 * c8 + 16>(xr[1][c8][1], x1);
      });
      lrx_wait_all<NCH>(xr[0]);
      lrx_wait_all<NCH>(xr[1]);
#pragma unroll
      for (int ch = 0; ch < NCH; ch++)
#pragma unroll
        for (int st = 0; st < 2; st++) {
          const f32x4 p0 = xr[st][ch][0], p1 = xr[st][ch][1];
          const float sxs = st ? sx1 : sx0;
          unsigned h0, h1, h2, h3, l0, l1, l2, l3;
          lr_split(p0.x * sxs, p0.y * sxs, h0, l0);
          lr_split(p0.z * sxs, p0.w * sxs, h1, l1);
          lr_split(p1.x * sxs, p1.y * sxs, h2, l2);
          lr_split(p1.z * sxs, p1.w * sxs, h3, l3);
          th[ch][st] = lrw_pack(h0, h1, h2, h3);
          tl[ch][st] = lrw_pack(l0, l1, l2, l3);
        }
    }
    // ---- 2. T sx^2 = X_nnz^T X_nnz on the matrix cores, rows by lane swaps (see the implicit kernel) ----
    float rl[NS];
    {
      f32x16 t[2][2];
#pragma unroll
      for (int e = 0; e < 16; e++) t[0][0][e] = t[0][1][e] = t[1][0][e] = t[1][1][e] = 0.f;
#pragma unroll
      for (int ch = 0; ch < NCH; ch++) {
        const f16x8 a0h = th[ch][0], a0l = tl[ch][0], a1h = th[ch][1], a1l = tl[ch][1];
        t[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, a0h, t[0][0], 0, 0, 0);
        if constexpr (SL == 64) t[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, a0h, t[1][0], 0, 0, 0);
        if constexpr (SL == 64) t[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, a1h, t[0][1], 0, 0, 0);
        t[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, a1h, t[1][1], 0, 0, 0);
        t[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, a0l, t[0][0], 0, 0, 0);
        if constexpr (SL == 64) t[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, a0l, t[1][0], 0, 0, 0);
        if constexpr (SL == 64) t[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, a1l, t[0][1], 0, 0, 0);
        t[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, a1l, t[1][1], 0, 0, 0);
        t[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, a0h, t[0][0], 0, 0, 0);
        if constexpr (SL == 64) t[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, a0h, t[1][0], 0, 0, 0);
        if constexpr (SL == 64) t[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, a1h, t[0][1], 0, 0, 0);
        t[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, a1h, t[1][1], 0, 0, 0);
      }
      auto swp = [&](const int ta, const int v, float& c_lo, float& c_hi) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(t[ta][0][v]), __float_as_uint(t[ta][1][v]), false, false);
        c_lo = __uint_as_float(sw[0]);
        c_hi = __uint_as_float(sw[1]);
      };
      if constexpr (SL == 64) {
#pragma unroll
        for (int ta = 0; ta < 2; ta++)
#pragma unroll
          for (int v = 0; v < 16; v++) {
            const int c0 = 32 * ta + 8 * (v >> 2) + (v & 3);
            float y0, y1;
            swp(ta, v, y0, y1);
            if (c0 < NS) rl[c0] = y0;
            if (c0 + 4 < NS) rl[c0 + 4] = y1;
          }
      } else if constexpr (SL == 32) {
#pragma unroll
        for (int v = 0; v < 16; v++) {
          const int c0 = 8 * (v >> 2) + (v & 3);
          float a0, a1, b0, b1;
          swp(0, v, a0, a1);
          swp(1, v, b0, b1);
          rl[c0] = hf ? b0 : a0;
          rl[c0 + 4] = hf ? b1 : a1;
        }
      } else {
        const int gq = (ln >> 4) & 1;
#pragma unroll
        for (int v = 0; v < 8; v++) {
          const int c0 = 8 * (v >> 2) + (v & 3);
          float a0, a1, b0, b1, c0v, c1v, d0, d1;
          swp(0, v, a0, a1);
          swp(0, v + 8, b0, b1);
          swp(1, v, c0v, c1v);
          swp(1, v + 8, d0, d1);
          const float lo0 = gq ? b0 : a0, lo1 = gq ? b1 : a1, hi0 = gq ? d0 : c0v, hi1 = gq ? d1 : c1v;
          rl[c0] = hf ? hi0 : lo0;
          rl[c0 + 4] = hf ? hi1 : lo1;
        }
      }
    }
    // ---- 3. (T + lambda I) z = r.  T is taken out of the operands' scale first (two exact multiplications per entry; through round
    // 6 the system was solved in the scaled form (T sx^2 + lambda sx^2 I) z' = r, and with factors that have shrunk far enough --
    // lambda = 1000 of the reference's test grid takes them to 1e-28 in five iterations -- lambda sx^2 overflowed, z' = 0, and every
    // such row came back as zeros: tools/dbg/chol_lambda1000_fit.py, native ranks 64 and 128).  A T that underflows here is below
    // 2^-126 against lambda; powers of two commute with every operation of the solve, so nothing else changes by a bit. ----
    float z = 0.f, dummy = 0.f;
#pragma unroll
    for (int t = 0; t < NS; t++) rl[t] = (rl[t] * ux) * ux;
    lrw_solve<SL, NS>(rl, rating, ln, false, lam_i, z, dummy);
    const float e_i = valid ? z : 0.f;
    const float res = lam_i * e_i;   // r_j - x_j . y
    const float loss_i = valid ? res * res : 0.f;
    asm volatile("" : "+v"(id_nx), "+v"(c_nx), "+v"(p1_nn), "+v"(n_nn), "+v"(rid_n3));   // (see the implicit kernel)
    wave_sync();
    sE[ln] = e_i;
    wave_sync();
    // ---- 4. y = sum_j z_j x_j: the terms times the identity turn the vectors to lane = coordinate ----
    float yreg = 0.f;   // sum over the pass's rows of lambda_use |y|^2
    {
      float4 ec[2][4];
#pragma unroll
      for (int st = 0; st < 2; st++)
#pragma unroll
        for (int q = 0; q < 4; q++) ec[st][q] = *reinterpret_cast<const float4*>(sE + 32 * st + 8 * q + 4 * hf);
      lr_sfor<NB>([&](auto cbt) {
        constexpr int cb = decltype(cbt)::value;
        f32x16 p0, p1;
#pragma unroll
        for (int e = 0; e < 16; e++) p0[e] = p1[e] = 0.f;
#pragma unroll
        for (int cc = 0; cc < 2; cc++) {
          p0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(th[2 * cb + cc][0], idf[cc], p0, 0, 0, 0);
          p1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(th[2 * cb + cc][1], idf[cc], p1, 0, 0, 0);
          p0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(tl[2 * cb + cc][0], idf[cc], p0, 0, 0, 0);
          p1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(tl[2 * cb + cc][1], idf[cc], p1, 0, 0, 0);
        }
        float q[4];
#pragma unroll
        for (int a2 = 0; a2 < 4; a2++) {
          const float4 e0 = ec[0][a2], e1 = ec[1][a2];
          const float s0 = fmaf(e0.x, p0[4 * a2], fmaf(e0.y, p0[4 * a2 + 1], fmaf(e0.z, p0[4 * a2 + 2], e0.w * p0[4 * a2 + 3])));
          const float s1 = fmaf(e1.x, p1[4 * a2], fmaf(e1.y, p1[4 * a2 + 1], fmaf(e1.z, p1[4 * a2 + 2], e1.w * p1[4 * a2 + 3])));
          if (a2 & 1) { q[a2 >> 1] += s0; q[2 + (a2 >> 1)] += s1; }
          else { q[a2 >> 1] = s0; q[2 + (a2 >> 1)] = s1; }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const unsigned uq = __float_as_uint(q[r]);
          const auto sw = __builtin_amdgcn_permlane32_swap(uq, uq, false, false);
          q[r] = (__uint_as_float(sw[0]) + __uint_as_float(sw[1])) * ux;
        }
        float yr[4];
        if constexpr (SL == 64) { yr[0] = (q[0] + q[1]) + (q[2] + q[3]); yr[1] = yr[2] = yr[3] = 0.f; }
        else if constexpr (SL == 32) { yr[0] = q[0] + q[1]; yr[1] = q[2] + q[3]; yr[2] = yr[3] = 0.f; }
        else { yr[0] = q[0]; yr[1] = q[1]; yr[2] = q[2]; yr[3] = q[3]; }
#pragma unroll
        for (int r = 0; r < 4; r++) {
          if (r < (64 >> lsh)) {   // wave-uniform
            const int rid = __builtin_amdgcn_readlane(rid_c, r << lsh);
            const float lam_r = readlane_f(lam_i, r << lsh);
            if (rid >= 0 && hf == 0) {
              a.Y[(size_t)rid * KP + 32 * cb + n] = yr[r];
              yreg = fmaf(lam_r * yr[r], yr[r], yreg);
            }
          }
        }
      });
    }
    const float lsum = wave_sum(loss_i), ysum = wave_sum(yreg);
    wloss += (double)(lsum + ysum);
    rid_c = rid_n; n_c = n_n;
    rid_n = rid_nn; p1_n = p1_nn; n_n = n_nn;
    rid_nn = rid_n3;
    id_c = id_nx; c_c = c_nx;
  }
  if (lane == 0) sRed[wv] = wloss;
  __syncthreads();
  if (tid == 0) {
    double s2 = 0.0;
    for (int w = 0; w < 8; w++) s2 += sRed[w];
    a.loss_partials[loss_slot0 + blockIdx.x] = s2;
  }
}

}  // namespace

// dev builds (-DRSP_AB): RSPARSE_HIP_LR_WAVE=0 keeps rank 128 on the workgroup kernel, for same-box comparisons
static bool lr_wave_on() {
#ifdef RSP_AB
  static const bool on = [] {
    const char* e = std::getenv("RSPARSE_HIP_LR_WAVE");
    return !(e && e[0] == '0');
  }();
  return on;
#else
  return true;
#endif
}

bool chol_lr_supported(const AlsArgs& a, bool implicit) {
  return implicit && a.k > 96 && a.k <= 128 && a.k % 2 == 0 && !a.rhs_vals && !a.loss_tgt && !a.rhs_init &&
         (reinterpret_cast<uintptr_t>(a.X) & 7) == 0;
}

// rows: the n_rows rows of 1..kCholLrMax non-zeros (a suffix of the length-sorted order); M / Mt: 2 x 128 x 128 floats of
// scratch (M: the two fp16 terms of Mt, Mt: fp32); flags: word 2 of the statistics block launch_ne_stats leaves (word 0 =
// bits of max |X| -- X must have been scanned --, word 2 = some confidence < 1, word 3 = the prep kernel's exponent) -- the
// prep kernel ORs its own verdict into word 2.  Loss partials of its kCholLrGrid workgroups from loss_slot0 on.
hipError_t launch_als_chol_lr(const AlsArgs& a, const int32_t* rows, int n_rows, float* M, float* Mt, unsigned* flags,
                              int loss_slot0, hipStream_t s, hipEvent_t* ev_slot) {
  // the list is longest first: lr_n_gt32 rows of 33..64 non-zeros, then (lr_n_gt16 - lr_n_gt32) of 17..32, then the rest;
  // without the counts every row is a pass of its own
  int n64 = n_rows, n32 = 0;
  if (a.lr_n_gt32 >= 0 && a.lr_n_gt16 >= a.lr_n_gt32 && a.lr_n_gt16 <= n_rows) {
    n64 = a.lr_n_gt32;
    n32 = a.lr_n_gt16 - a.lr_n_gt32;
  }
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
  _Float16* MT16 = reinterpret_cast<_Float16*>(Mt + 128 * 128);   // (the scratch is 3 x 128 x 128 floats)
  const bool wave_form = lr_wave_on() && a.k == KP;
  hipLaunchKernelGGL(prep, dim3(1), dim3(256), prep_lds, s, a.XtX, a.k, reinterpret_cast<_Float16*>(M), Mt, flags,
                     wave_form ? MT16 : nullptr);
  if ((err = hipGetLastError()) != hipSuccess) return err;
  if (wave_form) {   // rank 128: one wave per pass, one launch per class of passes
    int n48 = n64;   // rows of more than 48 non-zeros (a prefix of the n64 rows of 33..64)
    if (a.lr_n_gt48 >= 0 && a.lr_n_gt48 <= n64) n48 = a.lr_n_gt48;
    const int P64 = n64, P32 = (n32 + 1) / 2, P16 = (n_rows - n64 - n32 + 3) / 4;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    cus = std::min(cus, kCholLrGrid / 4);
    auto go = [&](auto kw, const int lo, const int hi, const int slot, const bool note) -> hipError_t {
      if (hi <= lo) return hipSuccess;
      hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(kw), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)LrwSmem::bytes);
      if (e2 != hipSuccess) return e2;
      const int gridw = std::min((hi - lo + 7) / 8, cus);   // one resident workgroup of 8 waves per CU
      if (note) prof_note(ev_slot, reinterpret_cast<const void*>(kw));
      hipLaunchKernelGGL(kw, dim3(gridw), dim3(512), LrwSmem::bytes, s, a, rows, n_rows, n64, n32, lo, hi,
                         reinterpret_cast<const _Float16*>(M), reinterpret_cast<const _Float16*>(MT16), flags,
                         loss_slot0 + slot * cus);
      return hipGetLastError();
    };
    // (the event segment is named after the class with the most passes on the bench matrix)
    if ((err = go(als_chol_lrw_kernel<KP, 64, 64>, 0, n48, 0, false)) != hipSuccess) return err;
    if ((err = go(als_chol_lrw_kernel<KP, 64, 48>, n48, P64, 1, true)) != hipSuccess) return err;
    if ((err = go(als_chol_lrw_kernel<KP, 32, 32>, P64, P64 + P32, 2, false)) != hipSuccess) return err;
    return go(als_chol_lrw_kernel<KP, 16, 16>, P64 + P32, P64 + P32 + P16, 3, false);
  }
  auto kern = als_chol_lr_kernel<KP>;
  if ((err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)LrSmem<KP>::bytes)) != hipSuccess)
    return err;
  const int n_pass = n64 + (n32 + 1) / 2 + (n_rows - n64 - n32 + 3) / 4;
  const int grid = n_pass < kCholLrGrid ? n_pass : kCholLrGrid;
  prof_note(ev_slot, reinterpret_cast<const void*>(kern));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LrSmem<KP>::bytes, s, a, rows, n_rows, n64, n32,
                     reinterpret_cast<const _Float16*>(M), Mt, flags, loss_slot0);
  return hipGetLastError();
}


// explicit feedback, ranks 64 and 128, no bias operands, lambda > 0 (with the reference's default lambda = 0 the k x k system of a
// row of n < k ratings is singular: that case keeps the k x k kernels and their general-solver fall-back, which report it): the
// rows of 1..kCholLrMax ratings in push-through form
bool chol_lrx_supported(const AlsArgs& a, bool implicit) {
  return !implicit && (a.k == 64 || a.k == 128) && !a.rhs_vals && !a.loss_tgt && !a.rhs_init && a.lambda > 0.f &&
         (reinterpret_cast<uintptr_t>(a.X) & 15) == 0;
}

// rows / counts as launch_als_chol_lr; stats: word 0 = bits of max |X| (launch_ne_stats).  Loss partials from loss_slot0 on
// (kCholLrGrid slots, zeroed here).
hipError_t launch_als_chol_lrx(const AlsArgs& a, const int32_t* rows, int n_rows, const unsigned* stats, int loss_slot0,
                               hipStream_t s, hipEvent_t* ev_slot) {
  int n64 = n_rows, n32 = 0;
  if (a.lr_n_gt32 >= 0 && a.lr_n_gt16 >= a.lr_n_gt32 && a.lr_n_gt16 <= n_rows) {
    n64 = a.lr_n_gt32;
    n32 = a.lr_n_gt16 - a.lr_n_gt32;
  }
  hipError_t err;
  if ((err = hipMemsetAsync(a.loss_partials + loss_slot0, 0, (size_t)kCholLrGrid * sizeof(double), s)) != hipSuccess)
    return err;
  if (n_rows <= 0) return hipSuccess;
  int n48 = n64;
  if (a.lr_n_gt48 >= 0 && a.lr_n_gt48 <= n64) n48 = a.lr_n_gt48;
  const int P64 = n64, P32 = (n32 + 1) / 2, P16 = (n_rows - n64 - n32 + 3) / 4;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  cus = std::min(cus, kCholLrGrid / 4);
  auto go = [&](auto kw, const int lo, const int hi, const int slot, const bool note) -> hipError_t {
    if (hi <= lo) return hipSuccess;
    const int gridw = std::min((hi - lo + 7) / 8, cus);
    if (note) prof_note(ev_slot, reinterpret_cast<const void*>(kw));
    hipLaunchKernelGGL(kw, dim3(gridw), dim3(512), 0, s, a, rows, n_rows, n64, n32, lo, hi, stats, loss_slot0 + slot * cus);
    return hipGetLastError();
  };
#define RSP_LRX(KPV)                                                                                        \
  {                                                                                                         \
    if ((err = go(als_chol_lrx_kernel<KPV, 64, 64>, 0, n48, 0, false)) != hipSuccess) return err;           \
    if ((err = go(als_chol_lrx_kernel<KPV, 64, 48>, n48, P64, 1, true)) != hipSuccess) return err;          \
    if ((err = go(als_chol_lrx_kernel<KPV, 32, 32>, P64, P64 + P32, 2, false)) != hipSuccess) return err;   \
    return go(als_chol_lrx_kernel<KPV, 16, 16>, P64 + P32, P64 + P32 + P16, 3, false);                      \
  }
  if (a.k == 64) RSP_LRX(64)
  if (a.k == 128) RSP_LRX(128)
#undef RSP_LRX
  return hipErrorInvalidValue;
}

}  // namespace rsparse_hip

#ifdef RSP_LRW_PROF
// dev builds: out[4][8] = per class (49..64, 33..48, 17..32, <= 16 non-zeros) the s_memtime ticks of the phases (issue of the
// gather, wait for it, split + V' GEMM, terms + T GEMM + lane swaps, n x n solve, P GEMM + y, bookkeeping) and the passes
extern "C" int rsparse_hip_dev_lrw_prof(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rsparse_hip::g_lrw_prof), sizeof(unsigned long long) * 32) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(rsparse_hip::g_lrw_prof), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#endif
