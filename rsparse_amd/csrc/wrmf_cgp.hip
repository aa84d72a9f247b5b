// CG half-iteration for the SHORTEST rows (1..16 non-zeros, and empty ones), two rows per wave (gfx950, wave64).
//
// Same arithmetic as als_cgq_kernel (wrmf_cgq.hip: cg_solver_implicit<T>, inst/include/wrmf_implicit.hpp:8-32, column loop
// :160-283; with a global bias cg_solver_implicit_global_bias, :35-57).  Half of the rows of the <= 32 bucket of the bench
// matrix have at most 16 non-zeros; in the one-wave kernel such a row fills 4 of the wave's 8 quads and still pays the
// whole fixed cost of a CG sweep (the shared dense product and its two barriers, the 16-lane and cross-group reductions,
// the scalar chain of alpha and beta): the kernel is bound by VALU issue, not by HBM (profiles/README.md).  Here a wave
// solves TWO rows side by side: lanes 0..31 one row, lanes 32..63 the other, every instruction serves both.
//
// Layout: a rank-128 vector is spread over the 16 lanes of a DPP row exactly as in wrmf_cgq.hip (8 floats per lane: floats
// [4i, 4i+4) and [64+4i, 64+4i+4) for lane i); a half-wave is two such groups, which hold two DIFFERENT non-zeros of the
// half's row at a time: slot s = 2 q + (group & 1), q = 0..7 -> 16 non-zeros in 64 registers per lane.  CG state is
// replicated in the two groups of a half and differs between the halves; all per-row scalars (row id, pointers, alpha,
// beta, convergence) are per-lane values that are uniform inside a half.  The dense product G v is shared by the
// workgroup's EIGHT rows on the matrix cores (G as two fp16 terms in registers, the vectors published as two fp16 terms
// through LDS, three products of order < 2: the DMF scheme of wrmf_cgq.hip with 8 instead of 4 live columns of the
// 16-column tile).  Slots beyond a row's length read the all-zero row and carry c = 0.
#include "wrmf_internal.h"
#include "wrmf_device.h"

namespace rsparse_hip {
namespace {

using namespace dev;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr float kCgTolP = 1e-10f;  // CG_TOL, inst/include/wrmf.hpp:22
constexpr int kPairMaxSweeps = 4;

__device__ __forceinline__ float p_row16_sum(float v) {
  v += dpp<0xB1>(v);
  v += dpp<0x4E>(v);
  v += dpp<0x141>(v);
  v += dpp<0x140>(v);
  return v;
}
__device__ __forceinline__ float p_row16_max(float v) {
  v = fmaxf(v, dpp<0xB1>(v));
  v = fmaxf(v, dpp<0x4E>(v));
  v = fmaxf(v, dpp<0x141>(v));
  v = fmaxf(v, dpp<0x140>(v));
  return v;
}
// sum over the two groups of a half-wave (lanes l and l ^ 16), result in both, bitwise identical
__device__ __forceinline__ float p_pair_sum(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ int p_scale_exp(float vmax) {
  const int eb = (int)((__float_as_uint(vmax) >> 23) & 0xffu);
  return min(253, max(1, 267 - eb));
}
__device__ __forceinline__ void p_split(const float x0, const float x1, unsigned& hi, unsigned& lo) {
  const f32x2 v = {x0, x1};
  const f16x2 h = __builtin_convertvector(v, f16x2);
  const f32x2 r = v - __builtin_convertvector(h, f32x2);
  const f16x2 l = __builtin_convertvector(r, f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}

struct PairSmem {
  static constexpr int KP = 128, ps = KP + 8, os = KP + 4;
  static constexpr size_t bytes = (size_t)2 * 8 * ps * 2 + (size_t)8 * os * 4 + (size_t)4 * 64 * 4 + 16;
};

// KFULL (rank 128): unconditional vector loads and the warm start requested before them -- the first sweep's publish / dense
// product run while the vectors are in flight and its quad pass takes them as they arrive (als_cgq_kernel's KFULL)
template <bool GB, bool KFULL = false>
__global__ __launch_bounds__(256, 2) void als_cgp_kernel(AlsArgs a, const int32_t* __restrict__ rows, int n_rows, int iters,
                                                        size_t loss_slot0) {
  constexpr int KP = 128, RPN = 8, VW = 4, NV = 2, NQ = 8;
  using SM = PairSmem;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  _Float16* sPh = reinterpret_cast<_Float16*>(smem);            // [8][KP + 8] leading fp16 terms of the published vectors
  _Float16* sPl = sPh + 8 * SM::ps;                             // [8][KP + 8] second terms
  float* sOut = reinterpret_cast<float*>(sPl + 8 * SM::ps);     // [8][KP + 4] G v (times the scales)
  float* sTsv = sOut + 8 * SM::os;                              // [4][2][32]  x_j . y accumulated / of the current step
  const int tid = threadIdx.x, lane = tid & 63, wv = rfl(tid >> 6);
  const int g = lane >> 4, i = lane & 15, H = lane >> 5, g2 = g & 1;
  const int k = KFULL ? KP : a.k;
  const float gbias = GB ? a.gbias : 0.f, ltgt = GB ? a.loss_tgt_const : 1.f;
  for (int e = tid; e < 4 * 64; e += 256) sTsv[e] = 0.f;
  // this wave's 32 rows of G as A operands of v_mfma_f32_16x16x32_f16 (tile t = rows 32 wv + 16 t + (lane & 15), step ks =
  // columns 32 ks + 8 (lane >> 4) + 0..7), two fp16 terms of G * 2^ge; ginv = 2^-ge
  f16x8 gAh[2][4], gAl[2][4];
  float ginv;
  {
    const int m = lane & 15, kb = lane >> 4;
    float gmax = 0.f;
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int ks = 0; ks < 4; ks++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const int r = 32 * wv + 16 * t + m, c = 32 * ks + 8 * kb + e;
          {   // (clamped address, select afterwards: a per-lane `cond ? load : 0` is a branch with a wait per element)
            // (... and a MULTIPLY by the 0 / 1 mask, not a select: hipcc sinks a load whose value only one side of a select uses
            //  back under the condition)
            const float gv = a.XtX[(size_t)min(r, k - 1) * k + min(c, k - 1)];
            gmax = fmaxf(gmax, fabsf(gv) * ((r < k && c < k) ? 1.f : 0.f));
          }
        }
    for (int o = 32; o > 0; o >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o));
    const int ge = p_scale_exp(gmax);
    const float gs = __uint_as_float((unsigned)ge << 23);
    ginv = __uint_as_float((unsigned)(254 - ge) << 23);
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        unsigned hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const int r = 32 * wv + 16 * t + m, c = 32 * ks + 8 * kb + e;
          const float g0v = a.XtX[(size_t)min(r, k - 1) * k + min(c, k - 1)];
          const float g0 = g0v * ((r < k && c < k) ? 1.f : 0.f);
          const float g1v = a.XtX[(size_t)min(r, k - 1) * k + min(c + 1, k - 1)];
          const float g1 = g1v * ((r < k && c + 1 < k) ? 1.f : 0.f);
          p_split(g0 * gs, g1 * gs, hi[e / 2], lo[e / 2]);
        }
        const uint4 h4 = {hi[0], hi[1], hi[2], hi[3]}, l4 = {lo[0], lo[1], lo[2], lo[3]};
        gAh[t][ks] = __builtin_bit_cast(f16x8, h4);
        gAl[t][ks] = __builtin_bit_cast(f16x8, l4);
      }
  }
  __syncthreads();
  float* tacc = sTsv + wv * 64 + 16 * H;   // this half's 16 slots: x_j . y accumulated over the CG steps
  float* tcur = tacc + 32;                 // ... x_j . p of the current step
  const int col = 2 * wv + H;              // this half's column of the shared dense product
  double wloss = 0.0;
  const int wave_global = blockIdx.x * 4 + wv, total_waves = gridDim.x * 4;
  auto ridx = [&](const int itn) { return (wave_global + itn * total_waves) * 2 + H; };   // position in `rows`, per half
  // row ids run two iterations ahead, pointers one ahead (per-lane values, uniform inside a half)
  int row_c = 0, p1_c = 0, p2_c = 0, row_n = 0;
  if (iters > 0 && ridx(0) < n_rows) {
    row_c = rows[ridx(0)];
    p1_c = a.col_ptrs[row_c];
    p2_c = a.col_ptrs[row_c + 1];
  }
  if (iters > 1 && ridx(1) < n_rows) row_n = rows[ridx(1)];

  // KFULL: the indices / values of a pair of rows are requested during the PREVIOUS pair's sweeps (the first pair's here): the
  // gather at the row switch is then one HBM round trip -- the vectors -- instead of two (what the one- and two-wave kernels of
  // wrmf_cgq.hip have done since round 2; this kernel sits at 238 registers: 17 more).  All eight
  // slots of a lane are requested whatever the rows' lengths (a slot outside its row reads entry 0 and is dropped).
  int idn[NQ];
  float cvn[NQ], cln = 0.f;
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    idn[q] = 0;
    cvn[q] = 0.f;
  }
  auto request_indices = [&](const bool valid, const int np1, const int ncnt) {
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int s = 2 * q + g2;
      const int j = (valid && s < ncnt) ? np1 + s : 0;
      idn[q] = a.row_idx[j];
      cvn[q] = a.vals[j];
    }
    cln = a.vals[(valid && i < ncnt) ? np1 + i : 0];
  };
  auto settle_indices = [&]() {   // a use the compiler cannot move: the wait for the request sits HERE
    asm volatile("" : "+v"(idn[0]), "+v"(idn[1]), "+v"(idn[2]), "+v"(idn[3]), "+v"(idn[4]), "+v"(idn[5]), "+v"(idn[6]), "+v"(idn[7]));
    asm volatile("" : "+v"(cvn[0]), "+v"(cvn[1]), "+v"(cvn[2]), "+v"(cvn[3]), "+v"(cvn[4]), "+v"(cvn[5]), "+v"(cvn[6]), "+v"(cvn[7]), "+v"(cln));
  };
  if constexpr (KFULL) {
    request_indices(iters > 0 && ridx(0) < n_rows, p1_c, p2_c - p1_c);
    settle_indices();   // (once per wave: the first pair pays the round trip)
  }

  for (int it = 0; it < iters; ++it) {
    const bool have = ridx(it) < n_rows;
    const int row = have ? row_c : 0, p1 = have ? p1_c : 0, p2 = have ? p2_c : 0;
    {
      int p1_n = 0, p2_n = 0, row_nn = 0;
      if (it + 1 < iters && ridx(it + 1) < n_rows) {
        p1_n = a.col_ptrs[row_n];
        p2_n = a.col_ptrs[row_n + 1];
      }
      if (it + 2 < iters && ridx(it + 2) < n_rows) row_nn = rows[ridx(it + 2)];
      row_c = row_n;
      p1_c = p1_n;
      p2_c = p2_n;
      row_n = row_nn;
    }
    const int cnt = p2 - p1;   // 0..16 (launcher)
    float* yrow = a.Y + (size_t)row * k;
    const bool live = have && (GB || cnt > 0);
    if (have && !live) {  // empty column -> zeros (wrmf_implicit.hpp:281); the 32 lanes of the half write it
      for (int e = lane & 31; e < k; e += 32) yrow[e] = 0.f;
    }
    // does any of the wave's two rows reach the second block of slots (8..15)?  (wave-uniform)
    const bool blk2 = max(__builtin_amdgcn_readlane(cnt, 0), __builtin_amdgcn_readlane(cnt, 32)) > 8;

    float x[RPN], r[RPN], p[RPN], ap[RPN];
    auto load_warm_start = [&]() {
#pragma unroll
      for (int b = 0; b < NV; b++) {
        const int off = b * 16 * VW + i * VW;
        float4 pc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (have && off < k) pc = *reinterpret_cast<const float4*>(yrow + off);  // warm start
        const float* pf = reinterpret_cast<const float*>(&pc);
#pragma unroll
        for (int c = 0; c < VW; c++) x[b * VW + c] = pf[c];
      }
    };
    if constexpr (KFULL) load_warm_start();
    // ---- gather: all index loads, then all vector loads; slots beyond the row read the zero row and carry c = 0 ----
    float xt[NQ][RPN], cv[NQ];
    {
      // (the loads WITHOUT a per-lane predicate: `in ? load : 0` compiles to a branch per slot with the address arithmetic of
      //  the vector load -- and its wait for the index -- inside: up to eight index round trips one after the other per pair of
      //  rows, found in the listing in round 5.  A slot outside its row reads entry 0 of the matrix and drops the result.)
      int id[NQ];
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        if constexpr (KFULL) {
          id[q] = idn[q];
          cv[q] = cvn[q];
        } else {
          const int s = 2 * q + g2;
          const int j = s < cnt ? p1 + s : 0;
          id[q] = 0;
          cv[q] = 0.f;
          if (q < 4 || blk2) {   // wave-uniform
            id[q] = a.row_idx[j];
            cv[q] = a.vals[j];
          }
        }
      }
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        const bool in = 2 * q + g2 < cnt;
        id[q] = in ? id[q] : 0;
        cv[q] = in ? cv[q] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        const bool in = 2 * q + g2 < cnt;
        const float* src = in ? a.X + (size_t)id[q] * k : a.zero_row;
#pragma unroll
        for (int b = 0; b < NV; b++) {
          const int off = b * 16 * VW + i * VW;
          if constexpr (KFULL) {
            // the first block without a branch; the second only when one of the wave's two rows reaches it (half of the rows of
            // this launch have at most 8 non-zeros: gathering the zero row for them cost 0.4 ms of the launch's 5.7)
            float4 pc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < 4 || blk2) pc = *reinterpret_cast<const float4*>(src + off);
            const float* pf = reinterpret_cast<const float*>(&pc);
#pragma unroll
            for (int c = 0; c < VW; c++) xt[q][b * VW + c] = pf[c];
          } else {
            float4 pc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < 4 || blk2) pc = *reinterpret_cast<const float4*>(src + min(off, k - VW));
            const float* pf = reinterpret_cast<const float*>(&pc);
#pragma unroll
            for (int c = 0; c < VW; c++) xt[q][b * VW + c] = off < k ? pf[c] : 0.f;
          }
        }
      }
    }
    // confidence of non-zero i of the half's row (loss)
    float cl;
    if constexpr (KFULL) cl = (g2 == 0 && i < cnt) ? cln : 0.f;
    else cl = (g2 == 0 && i < cnt) ? a.vals[p1 + i] : 0.f;

    if constexpr (!KFULL) load_warm_start();

    // mode 0: out = X_nnz (c - c1 % (X_nnz^T v + g)) - G v (+ base); mode 1: out = X_nnz (c1 % X_nnz^T v) + G v; mode 2: loss
    auto sweep = [&](const float(&v)[RPN], const int mode, float(&out)[RPN], float& loss_out) {
      float acc[RPN];
#pragma unroll
      for (int rr = 0; rr < RPN; rr++) acc[rr] = 0.f;
      float vinv = 0.f;
      if (mode != 2) {
        // (1) publish v as two fp16 terms of v * 2^ve: group g2 publishes piece g2 (elements [64 g2 + 4 i, + 4))
        float vmax = 0.f;
#pragma unroll
        for (int rr = 0; rr < RPN; rr++) vmax = fmaxf(vmax, fabsf(v[rr]));
        vmax = p_row16_max(vmax);
        const int ve = p_scale_exp(vmax);
        const float vs = __uint_as_float((unsigned)ve << 23);
        vinv = __uint_as_float((unsigned)(254 - ve) << 23) * (mode == 0 ? -1.f : 1.f);
        {
          float w4[VW];
#pragma unroll
          for (int c = 0; c < VW; c++) w4[c] = (g2 == 0 ? v[c] : v[VW + c]) * vs;
          unsigned h0, l0, h1, l1;
          p_split(w4[0], w4[1], h0, l0);
          p_split(w4[2], w4[3], h1, l1);
          const int off = col * SM::ps + g2 * 16 * VW + i * VW;
          *reinterpret_cast<uint2*>(sPh + off) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(sPl + off) = make_uint2(l0, l1);
        }
        __syncthreads();
        // (2) this wave's 32 rows of G against the eight vectors (columns 8..15 of the tile repeat them)
        f32x4 d0[2], d1[2];
#pragma unroll
        for (int t = 0; t < 2; t++) d0[t] = d1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nb = (lane & 7) * SM::ps + 8 * (lane >> 4);
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
          const f16x8 bh = *reinterpret_cast<const f16x8*>(sPh + nb + 32 * ks);
          const f16x8 bl = *reinterpret_cast<const f16x8*>(sPl + nb + 32 * ks);
#pragma unroll
          for (int t = 0; t < 2; t++) {
            d0[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gAh[t][ks], bh, d0[t], 0, 0, 0);
            d1[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gAh[t][ks], bl, d1[t], 0, 0, 0);
            d1[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gAl[t][ks], bh, d1[t], 0, 0, 0);
          }
        }
        if ((lane & 15) < 8) {   // D: column = lane & 15, rows 4 (lane >> 4) + 0..3 of the tile
#pragma unroll
          for (int t = 0; t < 2; t++) {
            const f32x4 d = (d0[t] + d1[t]) * ginv;
            *reinterpret_cast<f32x4*>(sOut + (lane & 15) * SM::os + 32 * wv + 16 * t + 4 * (lane >> 4)) = d;
          }
        }
      }
      float lacc = 0.f;
      if (mode != 2) {
        float* trec = (mode == 0 ? tacc : tcur) + g2;   // slot 2 q + g2 <- t (the 16 lanes of the group write the same value)
#pragma unroll
        for (int q0 = 0; q0 < NQ; q0 += 4) {
          if (q0 == 0 || blk2) {   // wave-uniform
            // (the four quads' dot chains and DPP reductions interleaved: see quad_pass in wrmf_cgq.hip)
            float t[4];
            f32x2 s2[4];
#pragma unroll
            for (int u = 0; u < 4; u++) s2[u] = f32x2{0.f, 0.f};
#pragma unroll
            for (int rr = 0; rr < RPN; rr += 2) {
              const f32x2 va = {v[rr], v[rr + 1]};
#pragma unroll
              for (int u = 0; u < 4; u++) {
                const f32x2 xa = {xt[q0 + u][rr], xt[q0 + u][rr + 1]};
                s2[u] = __builtin_elementwise_fma(xa, va, s2[u]);
              }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) t[u] = s2[u].x + s2[u].y;
#pragma unroll
            for (int u = 0; u < 4; u++) t[u] += dpp<0xB1>(t[u]);
#pragma unroll
            for (int u = 0; u < 4; u++) t[u] += dpp<0x4E>(t[u]);
#pragma unroll
            for (int u = 0; u < 4; u++) t[u] += dpp<0x141>(t[u]);
#pragma unroll
            for (int u = 0; u < 4; u++) t[u] += dpp<0x140>(t[u]);
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const int q = q0 + u;
              const float c = cv[q];
              trec[2 * q] = t[u];
              const float w = mode == 0 ? c - (c - 1.f) * (GB ? t[u] + gbias : t[u]) : (c - 1.f) * t[u];
#pragma unroll
              for (int rr = 0; rr < RPN; rr++) acc[rr] = fmaf(w, xt[q][rr], acc[rr]);
            }
          }
        }
      } else {
        // loss from t_acc = X_nnz^T y built up by the sweeps (the vectors are not touched again)
        wave_sync();
        const float t = tacc[i];
        const float dd = ltgt - t;
        lacc = p_pair_sum(p_row16_sum((g2 == 0 && i < cnt) ? cl * dd * dd : 0.f));
      }
      if (mode != 2) {
        // (3) the products are complete: fold this row's G v into group 0's partial sums, then reduce the two groups
        __syncthreads();
        const float f = g2 == 0 ? vinv : 0.f;
#pragma unroll
        for (int b = 0; b < NV; b++) {
          const f32x4 o = *reinterpret_cast<const f32x4*>(sOut + col * SM::os + b * 16 * VW + i * VW);
#pragma unroll
          for (int c = 0; c < VW; c++) acc[b * VW + c] = fmaf(f, o[c], acc[b * VW + c]);
        }
#pragma unroll
        for (int rr = 0; rr < RPN; rr++) out[rr] = p_pair_sum(acc[rr]);
        if constexpr (GB) {
          if (mode == 0) {
#pragma unroll
            for (int b = 0; b < NV; b++) {
              const int off = b * 16 * VW + i * VW;
              const float4 pc = *reinterpret_cast<const float4*>(a.rhs_init + min(off, k - VW));
              const float* pf = reinterpret_cast<const float*>(&pc);
#pragma unroll
              for (int c = 0; c < VW; c++) out[b * VW + c] += off < k ? pf[c] : 0.f;
            }
          }
        }
      } else {
        loss_out = lacc;
      }
    };
    auto dot16 = [&](const float(&u)[RPN], const float(&w)[RPN]) {
      float s = 0.f;
#pragma unroll
      for (int rr = 0; rr < RPN; rr++) s = fmaf(u[rr], w[rr], s);
      return p_row16_sum(s);
    };

    float dummy = 0.f;
    sweep(x, 0, r, dummy);
    if constexpr (KFULL)   // (p1_c / p2_c are the next pair's since the top of this iteration)
      request_indices(it + 1 < iters && ridx(it + 1) < n_rows, p1_c, p2_c - p1_c);
#pragma unroll
    for (int rr = 0; rr < RPN; rr++) p[rr] = r[rr];
    float rsold = dot16(r, r);
    bool conv = !live;   // a dead half keeps in step: its vectors are zero, its updates are masked
    for (int itc = 0; itc < a.cg_steps; ++itc) {
      sweep(p, 1, ap, dummy);
      const float pap = dot16(p, ap);
      // rsold / alpha / beta as the reference holds them: double scalars fed by T-valued dot products (wrmf_implicit.hpp:18-27)
      const float alpha = conv ? 0.f : (float)((double)rsold / (double)pap);
      wave_sync();
      if (g2 == 0) tacc[i] = fmaf(alpha, tcur[i], tacc[i]);   // X_nnz^T x += alpha X_nnz^T p
      wave_sync();
#pragma unroll
      for (int rr = 0; rr < RPN; rr++) {
        x[rr] = fmaf(alpha, p[rr], x[rr]);
        r[rr] = fmaf(-alpha, ap[rr], r[rr]);
      }
      const float rsnew = dot16(r, r);   // (outside the per-half branch: the DPP reductions want whole rows of lanes)
      if (!conv) {
        if (rsnew < kCgTolP) {
          conv = true;
        } else {
          const float beta = (float)((double)rsnew / (double)rsold);
#pragma unroll
          for (int rr = 0; rr < RPN; rr++) p[rr] = fmaf(p[rr], beta, r[rr]);
          rsold = rsnew;
        }
      }
    }
    float rl = 0.f;
    sweep(x, 2, ap, rl);
    if constexpr (KFULL) {
      // settle the prefetched indices / values here (requested four sweeps ago): left to the next pair's first use, the wait would
      // sit behind that pair's warm-start request and behind this pair's store -- the vectors would be requested a round trip late
      settle_indices();
    }
    if (live) {
      const float xx = dot16(x, x);
      wloss += (double)rl + a.lambda_loss * (double)xx;
      if (g2 == 0) {
#pragma unroll
        for (int b = 0; b < NV; b++) {
          const int off = b * 16 * VW + i * VW;
          if (off < k) {
            float4 pc;
            float* pf = reinterpret_cast<float*>(&pc);
#pragma unroll
            for (int c = 0; c < VW; c++) pf[c] = x[b * VW + c];
            *reinterpret_cast<float4*>(yrow + off) = pc;
          }
        }
      }
    }
  }
  const double other = __shfl(wloss, 32);
  if (lane == 0) a.loss_partials[loss_slot0 + (size_t)blockIdx.x * 4 + wv] = wloss + other;
}

}  // namespace

bool cgp_supported(int k, bool implicit) { return implicit && k > 64 && k <= 128 && k % 4 == 0; }

// grid (workgroups of 4 waves x 2 rows) for n_rows rows: about 32 pairs per wave, small sets spread over the CUs first
int cgp_grid(int n_rows) {
  if (n_rows <= 0) return 0;
  const long pairs = ((long)n_rows + 1) / 2;
  long per_wave = 32;
  const long spread = 4L * 512;
  if (pairs < spread * per_wave) per_wave = (pairs + spread - 1) / spread;
  if (per_wave < 1) per_wave = 1;
  return (int)((pairs + 4 * per_wave - 1) / (4 * per_wave));
}

// rows: n_rows row ids, each with at most 16 non-zeros (empty ones included), rank 65..128 (multiple of 4), implicit feedback;
// loss partials: cgp_grid(n_rows) * 4 slots from loss_slot0 on
hipError_t launch_als_cgp(const AlsArgs& a, const int32_t* rows, int n_rows, size_t loss_slot0, hipStream_t s,
                          hipEvent_t* ev_slot) {
  const int grid = cgp_grid(n_rows);
  if (grid <= 0) return hipSuccess;
  const long pairs = ((long)n_rows + 1) / 2;
  const int iters = (int)((pairs + (long)grid * 4 - 1) / ((long)grid * 4));
  const bool gb = a.gbias != 0.f;
  auto k0 = als_cgp_kernel<false, false>;
  auto k0f = als_cgp_kernel<false, true>;
  auto k1 = als_cgp_kernel<true, false>;
  const bool full = !gb && a.k == 128;
  const void* fn = gb ? reinterpret_cast<const void*>(k1) : (full ? reinterpret_cast<const void*>(k0f) : reinterpret_cast<const void*>(k0));
  hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PairSmem::bytes);
  if (err != hipSuccess) return err;
  (void)ev_slot;   // (the bucket's segment is named after its main kernel)
  if (gb) hipLaunchKernelGGL(k1, dim3(grid), dim3(256), PairSmem::bytes, s, a, rows, n_rows, iters, loss_slot0);
  else if (full) hipLaunchKernelGGL(k0f, dim3(grid), dim3(256), PairSmem::bytes, s, a, rows, n_rows, iters, loss_slot0);
  else hipLaunchKernelGGL(k0, dim3(grid), dim3(256), PairSmem::bytes, s, a, rows, n_rows, iters, loss_slot0);
  return hipGetLastError();
}

}  // namespace rsparse_hip
