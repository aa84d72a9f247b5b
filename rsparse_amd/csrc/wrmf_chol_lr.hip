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
// Needs every confidence >= 1 (D^1/2) and XtX positive definite: both are decided on the device (flags[0] != 0 ->
// this kernel returns at once and wrmf_chol.hip's kernel, which otherwise skips the short rows, takes them).
#include <type_traits>
#include <utility>

#include "wrmf_internal.h"
#include "wrmf_device.h"

namespace rsparse_hip {
namespace {

using namespace dev;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
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
constexpr int kLrLd = 130;   // LDS row stride (floats) of V' (fp32)
constexpr int kLrLh = 136;   // ... (halves) of the fp16 terms of X_nnz and of W: 16-byte aligned rows for the operand reads
constexpr int kLrLs = 65;    // ... of the n x n system

// ---- XtX = L L^T, M = L^-T, Mt = M^T = L^-1: one workgroup, once per half-iteration ----------------------------------
// M and Mt are written KP x KP, zero padded.  flags[0] |= 1 when XtX is not positive definite.
// Also written: the two fp16 terms of Mt * 2^e (the B operand of V' = X_nnz M: M[kk][col] = Mt[col][kk], 8 consecutive kk
// per lane) into M16 = [2][KP][KP] halves, and e (biased exponent) into flags[1].
template <int KP>
__global__ __launch_bounds__(256) void chol_lr_prep_kernel(const float* __restrict__ G, int k, _Float16* __restrict__ M16,
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
  }
}

// ---- the rows ----------------------------------------------------------------------------------------------------
template <int KP>
struct LrSmem {
  static constexpr int NP = 64;
  static constexpr size_t x_floats = (size_t)NP * kLrLh;   // fp16 terms of X_nnz ([2][NP][kLrLh] halves), then of W, then S (NP x kLrLs floats)
  static constexpr size_t v_floats = (size_t)NP * kLrLd;   // V'
  static constexpr size_t vec_floats = 4 * NP + 4 * KP + 64;
  static constexpr size_t bytes = (x_floats + v_floats + vec_floats) * 4 + 64;
};

template <int KP>
__global__ __launch_bounds__(256, 2) void als_chol_lr_kernel(AlsArgs a, const int32_t* __restrict__ rows, int n_rows,
                                                             const _Float16* __restrict__ M16, const float* __restrict__ Mt,
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
  float* sGv = sT + NP;                  // [KP] g
  float* sQv = sGv + KP;                 // [KP] q
  float* sY = sQv + KP;                  // [KP] y (two partial halves are added through sP)
  float* sP = sY + KP;                   // [KP] second half of y
  double* sRed = reinterpret_cast<double*>(sP + KP);   // [8]
  const int tid = threadIdx.x, lane = tid & 63, wv = rfl(tid >> 6);
  const int col = lane & 31, half = lane >> 5;
  const int k = a.k;
  if (flags[0] != 0) return;   // some confidence < 1 or XtX not positive definite: wrmf_chol.hip takes these rows
  // operand scales: max |X| from the statistics block in front of the flags (launch_ne_stats: flags = stats + 2), the
  // exponent of Mt from the prep kernel
  const int ex = lr_scale_exp(fmaxf(__uint_as_float(flags[-2]), 1e-30f)), eM = (int)flags[1];
  const float sx = lr_pow2(ex), inv_xm = lr_pow2(254 - ex) * lr_pow2(254 - eM);   // 2^-(ex - 127) * 2^-(eM - 127)

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
  for (int rot = 0; it < n_rows; it += G, rot++) {
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
      auto put = [&](const int u) {
        const int j = wv + 4 * u;
        unsigned hi = 0u, lo = 0u;
        if (j < n) lr_split(v[u].x * sx, v[u].y * sx, hi, lo);
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
    // 2b. the fp16 terms of W = D^1/2 V' * 2^ew over the X_nnz terms (thread t: row t / 4, 32 columns)
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
    // 5. S = I + W W^T (lower tiles) on the matrix cores from the fp16 terms of W, then -> sS (over those terms)
    __syncthreads();   // the terms of W are complete (and g, h above have been formed from V')
    {
      const int rt = wv == 0 ? 0 : 1, ct = wv == 2 ? 1 : 0;
      const bool mine = !(RSP_LR_ABL & 2) && wv < (nrt == 1 ? 1 : 3);
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
      if (mine) {
        const float inv_w2 = lr_pow2(254 - ew) * lr_pow2(254 - ew);
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int i = 32 * rt + (e & 3) + 8 * (e >> 2) + 4 * half, c2 = 32 * ct + col;
          sS[i * LS + c2] = (i == c2 ? 1.f : 0.f) + (acc[e] + acc2[e]) * inv_w2;
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
    // (the solving wave rotates from row to row, offset by the workgroup: the two workgroups of a CU would otherwise both
    // solve on the SIMD that holds their wave 0 while the other three idle)
    if (!(RSP_LR_ABL & 4) && wv == ((blockIdx.x + rot) & 3)) {
      int i = lane;
      asm volatile("" : "+v"(i));   // laundered per row: hipcc otherwise hoists the 64 load addresses below out of the row loop and spills them
      auto solve = [&](auto np_tag) {
        constexpr int NS = decltype(np_tag)::value;
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
              rows_to_all<(NS > 32 ? 4 : 2)>(r[j], rep);
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
// scratch (M: the two fp16 terms of Mt, Mt: fp32); flags: word 2 of the statistics block launch_ne_stats leaves (word 0 =
// bits of max |X| -- X must have been scanned --, word 2 = some confidence < 1, word 3 = the prep kernel's exponent) -- the
// prep kernel ORs its own verdict into word 2.  Loss partials of its kCholLrGrid workgroups from loss_slot0 on.
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
  hipLaunchKernelGGL(prep, dim3(1), dim3(256), prep_lds, s, a.XtX, a.k, reinterpret_cast<_Float16*>(M), Mt, flags);
  if ((err = hipGetLastError()) != hipSuccess) return err;
  auto kern = als_chol_lr_kernel<KP>;
  if ((err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)LrSmem<KP>::bytes)) != hipSuccess)
    return err;
  const int grid = n_rows < kCholLrGrid ? n_rows : kCholLrGrid;
  prof_note(ev_slot, reinterpret_cast<const void*>(kern));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LrSmem<KP>::bytes, s, a, rows, n_rows,
                     reinterpret_cast<const _Float16*>(M), Mt, flags, loss_slot0);
  return hipGetLastError();
}

}  // namespace rsparse_hip
