// CG half-iteration, register-resident "quad layout" (gfx950, wave64).
//
// Same arithmetic as als_cg_short/long_kernel (cg_solver_implicit<T>, inst/include/wrmf_implicit.hpp:8-32;
// cg_solver_explicit<T>, inst/include/wrmf_explicit.hpp:8-31; column loop :160-283 / :68-144), re-laid-out
// so that the gathered factor vectors of a row live in the REGISTER FILE instead of LDS.  The LDS-tile
// kernels fit only 4 waves per CU (64 KB Gramian + 17 KB tile per wave) and are latency-bound; a CU has
// 512 KB of VGPRs, so here 8 waves per CU each keep up to 64 gathered vectors (128 VGPRs) resident and LDS
// holds only the shared Gramian and ~1.5 KB per wave.
//
// Layout: a wave is 4 DPP rows ("groups") of 16 lanes.  A rank-KP vector is spread over the 16 lanes of a
// group, RPN = KP/16 floats per lane (k=128: floats [4i,4i+4) and [64+4i,64+4i+4) for lane i -> every load
// instruction reads whole 256-B half vectors).  The 4 groups hold 4 DIFFERENT non-zeros of the row at a
// time (a "quad"); CG state (x, r, p, Ap) is replicated in the 4 groups.  Then, per quad:
//     t_j  = x_j . v      RPN/2 packed FMAs + a 4-step DPP reduction inside the 16-lane row (no cross-row traffic)
//     acc += w_j * x_j    RPN/2 packed FMAs, w_j is already uniform inside the group (no readlane, no LDS)
// and once per sweep the 4 group partials are all-reduced (2 permlane swaps per register).  The dense
// G*v product is split over the groups by k-range and folded into the same all-reduce.  The loss needs
// t_j = x_j . y for the final y = x0 + sum_s alpha_s p_s, i.e. the dot products the sweeps already formed: they
// are accumulated per non-zero (LDS for resident rows, an HBM scratch for streamed rows) instead of a fifth pass.
//
// Rows longer than one wave's capacity (64 non-zeros = 128 VGPRs) are solved by teams of WPR = 2/4/8
// waves of one workgroup: every wave keeps its own chunk resident, partial vectors are combined through
// LDS with one barrier per CG sweep.  Rows beyond the workgroup's capacity are streamed (STREAM = 1
// instantiation): the first 16 non-zeros per wave stay in LDS, the rest is re-gathered in every sweep.
#include <algorithm>
#include <cstdlib>

#include "wrmf_internal.h"
#include "wrmf_device.h"

namespace rsparse_hip {
namespace {

using namespace dev;

typedef float f32x2 __attribute__((ext_vector_type(2)));

// tuning switches (resident kernels with CAPQ >= the value use the feature)
#ifndef RSP_ZPAD_MINCAPQ
#define RSP_ZPAD_MINCAPQ 0
#endif
#ifndef RSP_TSAVE_MINCAPQ
#define RSP_TSAVE_MINCAPQ 0
#endif
#ifndef RSP_GVFIRST_MINCAPQ
#define RSP_GVFIRST_MINCAPQ 16
#endif
// streamed rows: request the next chunk's indices / values (lane-major) while the current chunk is in flight -- one
// HBM round trip per chunk instead of two (-3 % on the streamed launch).  Implicit instantiations only: the explicit
// one is at 256 VGPRs and would spill
#ifndef RSP_STREAM_IDX_PREFETCH
#define RSP_STREAM_IDX_PREFETCH 1
#endif
// streamed rows: quads per wave of the row's prefix that stay in LDS across the sweeps (0 disables).  4 = one
// quad-pass block = 128 non-zeros per 8-wave team = 64 KB of LDS next to the 64 KB Gramian; 5 measured 1.5 % faster
// on the launch but needs the idle quads of the second block zeroed (they are stale registers otherwise)
// resident rows: prefetch the next row's indices / values (one per lane, 2 VGPRs) during the current row's sweeps, so
// the gather at the row switch is one HBM round trip instead of two.  Teams of up to this many waves use it: -4..6 %
// on the 1- and 2-wave kernels; the 4- and 8-wave kernels sit at 256 VGPRs and the two registers spill (+5 %)
#ifndef RSP_IDX_PREFETCH_MAXWPR
#define RSP_IDX_PREFETCH_MAXWPR 2
#endif
#ifndef RSP_STREAM_PREFIX_Q
#define RSP_STREAM_PREFIX_Q 4
#endif

constexpr float kCgTolQ = 1e-10f;  // CG_TOL, inst/include/wrmf.hpp:22
constexpr int kMaxSavedSweeps = 4;  // streamed rows keep the dot products of up to this many CG steps

template <int KP>
struct QG {
  static constexpr int RPN = KP / 16;            // floats of one vector per lane
  static constexpr int VW = RPN >= 4 ? 4 : RPN;  // floats per load piece
  static constexpr int NV = RPN / VW;            // pieces per vector per lane
  // position inside the vector of register r of lane i
  __device__ static __forceinline__ int elem(int i, int r) { return (r / VW) * (16 * VW) + i * VW + (r % VW); }
};

template <int VW>
struct Piece;
template <>
struct Piece<4> { using type = float4; };
template <>
struct Piece<2> { using type = float2; };

// sum over the 16 lanes of a DPP row; every lane of the row gets the result (needs full EXEC)
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp<0xB1>(v);   // quad_perm:[1,0,3,2]
  v += dpp<0x4E>(v);   // quad_perm:[2,3,0,1]
  v += dpp<0x141>(v);  // row_half_mirror
  v += dpp<0x140>(v);  // row_mirror
  return v;
}
// sum over the 4 groups (lanes l, l^16, l^32, l^48), result in all 4, bitwise identical everywhere.
// v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second,
// v_permlane32_swap the upper half of the first with the lower half of the second; fed two copies of v
// they leave {v_even, v_even | ...} and {v_odd, v_odd | ...}, whose sum is the pairwise all-reduce --
// pure VALU, no LDS round trip (ds_bpermute costs ~100+ cycles of latency per stage).
__device__ __forceinline__ float groups_sum(float v) {
  {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  return v;
}

// The same sum for FOUR registers at once, as a reduce-scatter: row g of the result holds the total of a_g.  Fed two
// DIFFERENT registers a lane swap is already the exchange step of both -- (a0, a1) -> rows (a0: 0+1, a1: 0+1, a0: 2+3, a1: 2+3)
// -- so four registers cost 3 swaps + 3 adds instead of 8 + 8 (+ 8 copies: the swap overwrites both operands).  The pairing
// is groups_sum's, (0 + 1) + (2 + 3): the totals are bit for bit the same.
__device__ __forceinline__ float groups_reduce_scatter4(const float a0, const float a1, const float a2, const float a3) {
  const auto s01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(a0), __float_as_uint(a1), false, false);
  const float c = __uint_as_float(s01[0]) + __uint_as_float(s01[1]);
  const auto s23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(a2), __float_as_uint(a3), false, false);
  const float d = __uint_as_float(s23[0]) + __uint_as_float(s23[1]);
  const auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(c), __float_as_uint(d), false, false);
  return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}
// ... and back: row g of `r` to every row of out[g]
__device__ __forceinline__ void groups_all_gather4(const float r, float& o0, float& o1, float& o2, float& o3) {
  const unsigned u = __float_as_uint(r);
  const auto h = __builtin_amdgcn_permlane32_swap(u, u, false, false);            // rows (0, 1, 0, 1), (2, 3, 2, 3)
  const auto lo = __builtin_amdgcn_permlane16_swap(h[0], h[0], false, false);     // rows (0, 0, 0, 0), (1, 1, 1, 1)
  const auto hi = __builtin_amdgcn_permlane16_swap(h[1], h[1], false, false);
  o0 = __uint_as_float(lo[0]);
  o1 = __uint_as_float(lo[1]);
  o2 = __uint_as_float(hi[0]);
  o3 = __uint_as_float(hi[1]);
}
#ifndef RSP_NO_GRS
#define RSP_GRS 1
#else
#define RSP_GRS 0
#endif

// ---- dense product on the matrix cores (DMF instantiation: one-wave rows of <= 32 non-zeros at rank 65..128) ----
// The four rows a workgroup solves side by side share every G v product: G is held in REGISTERS as two fp16 terms
// (wave w owns rows [32w, 32w + 32) as A operands of v_mfma_f32_16x16x32_f16), the four vectors are published to LDS as
// two fp16 terms each, and G v = (Gh + Gl)(vh + vl) is accumulated in fp32 from the three products of order < 2
// (2^-21 per product; both operands scaled by powers of two so that their largest entry lands in [2^13, 2^14)).
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// power of two that brings `vmax` into [2^13, 2^14), as the biased exponent (1..253, so that its inverse is normal too)
__device__ __forceinline__ int fp16_scale_exp(float vmax) {
  const int eb = (int)((__float_as_uint(vmax) >> 23) & 0xffu);
  return min(253, max(1, 267 - eb));
}
// x (already scaled) -> fl16(x), fl16(x - fl16(x)); the residual is exact in fp32
__device__ __forceinline__ void split_f16(const float x0, const float x1, unsigned& hi, unsigned& lo) {
  const f32x2 v = {x0, x1};
  const f16x2 h = __builtin_convertvector(v, f16x2);
  const f32x2 r = v - __builtin_convertvector(h, f32x2);
  const f16x2 l = __builtin_convertvector(r, f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp<0xB1>(v));
  v = fmaxf(v, dpp<0x4E>(v));
  v = fmaxf(v, dpp<0x141>(v));
  v = fmaxf(v, dpp<0x140>(v));
  return v;
}

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(reinterpret_cast<uintptr_t>(p));  // low 32 bits of a generic LDS pointer = LDS byte address
}
// LDS-DMA, one dword per lane: LDS destination = M0 + lane * 4 (wave-uniform base), source = each lane's own pointer.  Counts in
// vmcnt like a load; the compiler does not know about it, which is harmless as long as nothing is issued between it and
// the explicit wait that precedes the first read of its destination (older operations complete first)
__device__ __forceinline__ void dma4(const void* g, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(g), "s"(lds_base) : "memory", "m0");
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// DMF: 0 = dense product on the vector units (G in LDS as fp32), 1 = on the matrix cores with this wave's rows of G as fp16
// terms in REGISTERS (rows <= 32 non-zeros: 64 registers are free), 2 = the same with the fp16 terms in LDS, stored in
// fragment order (rows of 33..64 non-zeros: the gathered vectors take 128 registers, G's 64 KB of LDS hold the terms instead
// of the fp32 matrix: 16 conflict-free 16-byte reads per wave and sweep instead of 72 + 128 packed FMAs)
template <int KP, int CAPQ, int WAVES, int WPR, int STREAM, bool IMPLICIT, int DMF = 0>
struct QSmem {
  static constexpr size_t gram_floats = (IMPLICIT && DMF != 1) ? (size_t)KP * KP : 0;   // (DMF 2: 4 x 2 x 4 x 2 KB fragments)
  static constexpr size_t vec_floats = IMPLICIT ? (size_t)WAVES * KP : 0;
  static constexpr size_t red_floats = WPR > 1 ? (size_t)2 * WAVES * KP + 2 * WAVES : 0;
  // resident rows: per wave, t_acc[CAP] = x_j . y accumulated over the CG steps and t_cur[CAP] = x_j . p of the
  // current step (the loss is rebuilt from them instead of a fifth pass over the registers)
  static constexpr size_t tsv_floats = (STREAM && !(RSP_STREAM_IDX_PREFETCH && IMPLICIT)) ? 0 : (size_t)WAVES * 2 * CAPQ * 4;
  // streamed rows: the first RSP_STREAM_PREFIX_Q quads of every wave (gathered in the first sweep) stay in LDS,
  // so the other sweeps re-gather only the rest of the row
  static constexpr size_t pre_floats = STREAM ? (size_t)WAVES * RSP_STREAM_PREFIX_Q * 4 * KP : 0;
  // DMF: the published fp16 terms of the workgroup's vectors (row stride KP + 8 halves) and the products (KP + 4 floats):
  // the paddings shift consecutive rows by four banks, so the operand reads / result writes of four rows do not collide
  static constexpr int dmf_ps = KP + 8, dmf_os = KP + 4;
  static constexpr size_t dmf_floats = DMF ? (size_t)WAVES * dmf_ps + (size_t)WAVES * dmf_os : 0;
  // resident rows on teams of more than RSP_IDX_PREFETCH_MAXWPR waves: the next row's indices / values (one per lane) land here
  // by LDS-DMA during the current row's sweeps -- those kernels have no two registers to hold them (KFULL instantiations)
  static constexpr size_t pfx_floats = (!STREAM && WPR > RSP_IDX_PREFETCH_MAXWPR && CAPQ * 4 == 64) ? (size_t)WAVES * 2 * 64 : 0;
  static constexpr size_t bytes = (gram_floats + vec_floats + red_floats + tsv_floats + pre_floats + dmf_floats + pfx_floats) * 4 + 16;
};

// GB: implicit feedback with a global bias (cg_solver_implicit_global_bias, wrmf_implicit.hpp:35-57,203): the first
// residual is  X_nnz (c - c1 % (X_nnz^T x + global_bias)) - XtX x + global_bias_base  (a.gbias, a.rhs_init), every row is
// solved -- empty ones too (:178) -- and the loss compares x_j.y with 1 - global_bias (a.loss_tgt_const, :262-264).
// KFULL: the rank IS the padded rank (k == KP: ranks 32 / 64 / 128).  The gathered vectors are then the destination registers of
// unconditional loads -- no select behind them --, so nothing has to wait for the whole gather: the first sweep of a resident row
// starts on the quads that have arrived (the warm start is requested BEFORE the vectors) while the rest is still in flight.
template <int KP, int CAPQ, int WAVES, int WPR, int STREAM, bool IMPLICIT, int DMF = 0, bool GB = false, bool KFULL = false>
__global__ __launch_bounds__(WAVES * 64, 2) void als_cgq_kernel(AlsArgs a, const int32_t* __restrict__ rows, int n_rows,
                                                             int rows_per_team, size_t loss_slot0) {
  using G_ = QG<KP>;
  constexpr int RPN = G_::RPN, VW = G_::VW, NV = G_::NV, CAP = CAPQ * 4, TEAMS = WAVES / WPR;
  using piece_t = typename Piece<VW>::type;
  using SM = QSmem<KP, CAPQ, WAVES, WPR, STREAM, IMPLICIT, DMF>;
  static_assert(WAVES % WPR == 0, "teams must tile the workgroup");
  constexpr bool GRS = RSP_GRS && RPN == 8 && VW == 4;   // the four-registers-at-once group reductions (rank 65..128)
  auto groups_all_reduce8 = [](float(&a8)[RPN]) {
    if constexpr (RPN == 8) {
      const float r0 = groups_reduce_scatter4(a8[0], a8[1], a8[2], a8[3]);
      const float r1 = groups_reduce_scatter4(a8[4], a8[5], a8[6], a8[7]);
      groups_all_gather4(r0, a8[0], a8[1], a8[2], a8[3]);
      groups_all_gather4(r1, a8[4], a8[5], a8[6], a8[7]);
    }
  };
  static_assert(!DMF || (IMPLICIT && KP == 128 && WAVES == 4 && WPR == 1 && STREAM == 0), "DMF geometry");
  static_assert(DMF != 1 || CAPQ == 8, "register-resident G terms: only the 8-quad kernel has the 64 registers");
  static_assert(!GB || IMPLICIT, "the global bias of explicit feedback is removed from the data (R/model_WRMF.R:278-282)");
  const float gbias = GB ? a.gbias : 0.f, ltgt = GB ? a.loss_tgt_const : 1.f;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sG = reinterpret_cast<float*>(smem);
  float* sVec = sG + SM::gram_floats;
  float* sRed = sVec + SM::vec_floats;  // [2][WAVES][KP]
  float* sRedL = sRed + (WPR > 1 ? 2 * WAVES * KP : 0);              // [2][WAVES]
  float* sTsv = sVec + SM::vec_floats + SM::red_floats;              // [WAVES][2][CAP]
  float* sPre = sTsv + SM::tsv_floats;                               // [WAVES][PQ][4][KP]  (streamed kernels)
  float* sPfx = sPre + SM::pre_floats;                               // [WAVES][2][64]  (DMAPF; never next to the DMF areas)
  _Float16* sPh = reinterpret_cast<_Float16*>(sPre + SM::pre_floats);   // DMF: [WAVES][KP + 8] high fp16 terms
  _Float16* sPl = sPh + WAVES * SM::dmf_ps;                             //      [WAVES][KP + 8] low terms
  float* sOut = reinterpret_cast<float*>(sPl + WAVES * SM::dmf_ps);     //      [WAVES][KP + 4] G v (times the scales)

  const int tid = threadIdx.x, lane = tid & 63, wv = rfl(tid >> 6);
  const int g = lane >> 4, i = lane & 15;
  const int team = wv / WPR, tw = wv % WPR;
  const int k = KFULL ? KP : a.k;
  if constexpr (IMPLICIT && DMF == 0) {
    // (eight loads in flight per thread: one load, one wait, one store per trip was 32 L2 round trips per workgroup start)
    constexpr int GU = (KP * KP) % (8 * WAVES * 64) == 0 ? 8 : ((KP * KP) % (2 * WAVES * 64) == 0 ? 2 : 1);
    static_assert((KP * KP) % (GU * WAVES * 64) == 0, "the Gramian copy walks whole batches");
    for (int e0 = tid; e0 < KP * KP; e0 += GU * WAVES * 64) {
      float gv[GU];
#pragma unroll
      for (int u = 0; u < GU; u++) {
        const int e = e0 + u * WAVES * 64, r = e / KP, c = e % KP;
        gv[u] = a.XtX[(size_t)min(r, k - 1) * k + min(c, k - 1)];
      }
#pragma unroll
      for (int u = 0; u < GU; u++) {
        const int e = e0 + u * WAVES * 64, r = e / KP, c = e % KP;
        sG[e] = (r < k && c < k) ? gv[u] : 0.f;
      }
    }
  }
  for (int e = tid; e < (int)(SM::vec_floats + SM::red_floats + SM::tsv_floats); e += WAVES * 64) sVec[e] = 0.f;
  __syncthreads();

  // One-wave rows of <= 32 non-zeros are bound by the LDS traffic of the dense G v product (64 KB per row and sweep) and
  // use only 64 registers for their vectors: the first NRES of the 8 four-row slabs of G that a 16-lane group walks stay
  // in registers for the whole launch (32 registers per slab), the rest is read from LDS as before.
  constexpr int NRES = (IMPLICIT && CAPQ == 8 && WPR == 1 && !STREAM && KP == 128 && !DMF) ? 2 : 0;
  piece_t gres[NRES > 0 ? NRES : 1][4][NV];
  if constexpr (NRES > 0) {
#pragma unroll
    for (int rs = 0; rs < NRES; rs++)
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int b = 0; b < NV; b++)
          gres[rs][u][b] = *reinterpret_cast<const piece_t*>(sG + (4 * (g + 4 * rs) + u) * KP + b * 16 * VW + i * VW);
  }
  // DMF: this wave's 32 rows of G as A operands (tile t = rows 32 wv + 16 t + (lane & 15), step ks = columns
  // 32 ks + 8 (lane >> 4) + 0..7), two fp16 terms of G * 2^ge; ginv = 2^-ge
  f16x8 gAh[DMF == 1 ? 2 : 1][DMF == 1 ? 4 : 1], gAl[DMF == 1 ? 2 : 1][DMF == 1 ? 4 : 1];
  // DMF 2: the same fragments in LDS, [wave][tile][step][term] x 1 KB (lane l's 16 bytes at l * 16: conflict-free)
  char* sGf = reinterpret_cast<char*>(sG) + (size_t)wv * 16 * 1024;
  float ginv = 1.f;
  if constexpr (DMF) {
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
    const int ge = fp16_scale_exp(gmax);
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
          split_f16(g0 * gs, g1 * gs, hi[e / 2], lo[e / 2]);
        }
        const uint4 h4 = {hi[0], hi[1], hi[2], hi[3]}, l4 = {lo[0], lo[1], lo[2], lo[3]};
        if constexpr (DMF == 1) {
          gAh[t][ks] = __builtin_bit_cast(f16x8, h4);
          gAl[t][ks] = __builtin_bit_cast(f16x8, l4);
        } else {
          *reinterpret_cast<uint4*>(sGf + ((t * 4 + ks) * 2 + 0) * 1024 + lane * 16) = h4;
          *reinterpret_cast<uint4*>(sGf + ((t * 4 + ks) * 2 + 1) * 1024 + lane * 16) = l4;
        }
      }
    if constexpr (DMF == 2) wave_sync();   // (each wave reads back only what it wrote)
  }
  float* vec = sVec + wv * KP;
  float* tacc = sTsv + wv * 2 * CAP;  // resident rows only
  float* tcur = tacc + CAP;
  int buf = 0;
  double wloss = 0.0;
#ifdef RSP_CGQ_PROF   // dev builds: ticks per phase, summed over the waves of the 8-wave launch (wrmf_capi.cpp prints them)
  unsigned long long cq_t[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long cq_l = __builtin_amdgcn_s_memtime();
#define CQ_T(j) { const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); cq_t[j] += t1_ - cq_l; cq_l = t1_; }
#else
#define CQ_T(j)
#endif
  int pf_id = 0, pf_cnt = -1;   // next row's share of this wave, lane-major (RSP_IDX_PREFETCH)
  float pf_c = 0.f;
  const int team_global = blockIdx.x * TEAMS + team;
  const int total_teams = gridDim.x * TEAMS;

  // Row metadata runs two iterations ahead of the solve (row id) / one ahead (row pointers), so the
  // rows[] -> col_ptrs[] dependent loads are off the critical path.
  auto row_index = [&](int itn) { return team_global + itn * total_teams; };
  int row_c = 0, p1_c = 0, p2_c = 0, row_n = 0;
  if (rows_per_team > 0 && row_index(0) < n_rows) {
    row_c = rows[row_index(0)];
    p1_c = a.col_ptrs[row_c];
    p2_c = a.col_ptrs[row_c + 1];
  }
  if (rows_per_team > 1 && row_index(1) < n_rows) row_n = rows[row_index(1)];
  for (int it = 0; it < rows_per_team; ++it) {
    const int ri = row_index(it);
    const bool have = ri < n_rows;
    if (WPR == 1 && !DMF && !have) break;  // no barriers on this path: a wave may simply stop
    const int row = have ? rfl(row_c) : 0;
    const int p1 = have ? rfl(p1_c) : 0;
    const int p2 = have ? rfl(p2_c) : 0;
    {  // issue the loads for the next iterations now; they are consumed at the top of the next iteration
      int p1_n = 0, p2_n = 0, row_nn = 0;
      if (it + 1 < rows_per_team && row_index(it + 1) < n_rows) {
        p1_n = a.col_ptrs[row_n];
        p2_n = a.col_ptrs[row_n + 1];
      }
      if (it + 2 < rows_per_team && row_index(it + 2) < n_rows) row_nn = rows[row_index(it + 2)];
      row_c = row_n;
      p1_c = p1_n;
      p2_c = p2_n;
      row_n = row_nn;
    }
    const int cnt = p2 - p1;
    float* yrow = a.Y + (size_t)row * k;
    if (!GB && WPR == 1 && cnt <= 0) {  // empty column -> zeros (wrmf_implicit.hpp:281, wrmf_explicit.hpp:142)
      if (!DMF || have)
        for (int e = lane; e < k; e += 64) yrow[e] = 0.f;
      if constexpr (!DMF) continue;   // DMF: the workgroup's waves share the dense products, so this wave keeps in step
    }
    // streamed rows: the first PTEAM non-zeros are the LDS-resident prefix (PCAP per wave), the rest is streamed
    constexpr int PQ = STREAM ? RSP_STREAM_PREFIX_Q : 0, PCAP = PQ * 4, PTEAM = PCAP * WPR;
    const int pre = STREAM ? min(cnt, PTEAM) : 0;
    const int pw = STREAM ? max(0, min(PCAP, pre - tw * PCAP)) : 0;   // this wave's share of the prefix
    const int nchunks = (cnt - pre + CAP - 1) / CAP;
    // STREAM == 0: every row of this launch fits the team's resident capacity (bucket thresholds);
    // STREAM == 1: rows beyond it -- each wave re-gathers its chunks in every sweep.
    constexpr bool resident = STREAM == 0;
    const float lam_use =
        IMPLICIT ? 0.f : (float)(a.lambda_loss * (a.dynamic_lambda ? (double)(float)cnt : 1.0));

    // streamed rows only: scratch for the per-non-zero dot products of every sweep (see the loss pass)
    float* tscr = nullptr;
    float alph[kMaxSavedSweeps];
#pragma unroll
    for (int s2 = 0; s2 < kMaxSavedSweeps; s2++) alph[s2] = 0.f;
    if constexpr (STREAM == 1) {
      if (a.tscr && have && a.cg_steps <= kMaxSavedSweeps) tscr = a.tscr + a.stream_off[ri];
    }

    float xt[CAPQ][RPN];  // gathered vectors: quad q, group g holds non-zero 4q+g of the chunk
    float cv[CAPQ];       // its confidence / rating (uniform inside the group)
    constexpr int NSL = (CAP + 63) / 64;   // slots of the chunk per lane (2 only for the rank <= 64 geometry, CAP = 128)
    float cl[NSL];        // resident rows: confidence / rating of non-zero `lane` (+ 64) of the chunk (loss)
#pragma unroll
    for (int s2 = 0; s2 < NSL; s2++) cl[s2] = 0.f;
    int ccnt = 0;

    // Gather n (1..CAP) non-zeros starting at `base` into the registers: all index loads, then all vector
    // loads, with no control flow in between (two dependent HBM round trips per chunk; per-block branches
    // here cost ~40 VGPRs of PHI copies and spill).  Resident rows (gathered once): slots beyond n read the
    // all-zero row and carry c = 0, so t = 0 and every weight derived from (c, t) is 0 -- they drop out of all
    // sums without a select in the sweeps.  Streamed rows (gathered every sweep): clamped duplicates of the
    // last non-zero, masked by `valid` in quad_pass (cheaper than the pointer selects per gather).
    constexpr bool ZPAD = STREAM == 0 && CAPQ >= RSP_ZPAD_MINCAPQ;
    constexpr bool TSAVE = STREAM == 0 && CAPQ >= RSP_TSAVE_MINCAPQ;
    constexpr bool GVFIRST = STREAM == 0 && CAPQ >= RSP_GVFIRST_MINCAPQ;
    constexpr bool IDXPF = CAP <= 64 && ((STREAM == 0 && WPR <= RSP_IDX_PREFETCH_MAXWPR) || (STREAM == 1 && IMPLICIT && RSP_STREAM_IDX_PREFETCH));
    // the LDS-DMA variant of the same prefetch for the kernels that have no registers for it (see QSmem::pfx_floats)
#ifdef RSP_NO_DMAPF   // dev builds: A/B of the LDS-DMA prefetch alone
    constexpr bool DMAPF = false;
#else
    constexpr bool DMAPF = KFULL && !IDXPF && SM::pfx_floats > 0;
#endif
    int* pfxI = reinterpret_cast<int*>(sPfx) + wv * 128;
    float* pfxC = sPfx + wv * 128 + 64;
    int pf_pos = -1;   // streamed rows: chunk the prefetch registers belong to
    auto gather_q = [&](auto nq_tag, const int base, const int n, const bool from_pf = false) {
      constexpr int NQG = decltype(nq_tag)::value;
      int id[NQG > 0 ? NQG : 1];
      if constexpr (DMAPF) {
        // one path: the indices / values come from the wave's LDS slot, where the previous row's sweeps left them; a row
        // that was not announced (the team's first) fetches them the same way and waits
        if (!from_pf && n > 0) {
          const int j = base + min(lane, n - 1);
          dma4(a.row_idx + j, lds_addr(pfxI));
          dma4(a.vals + j, lds_addr(pfxC));
          wait_vm0();
        }
        wave_sync();
#pragma unroll
        for (int q = 0; q < NQG; q++) {
          const int j = max(min(4 * q + g, n - 1), 0);
          id[q] = pfxI[j];
          cv[q] = (!ZPAD || 4 * q + g < n) ? pfxC[j] : 0.f;
        }
        if constexpr (TSAVE) cl[0] = lane < n ? pfxC[lane] : 0.f;
        wave_sync();
      } else if (IDXPF && from_pf) {
        // lane-major prefetch registers -> quad layout through the wave's t-slots (dead between two rows)
        int* xi = reinterpret_cast<int*>(tacc);
        wave_sync();
        xi[lane & (CAP - 1)] = pf_id;
        tcur[lane & (CAP - 1)] = pf_c;
        wave_sync();
#pragma unroll
        for (int q = 0; q < NQG; q++) {
          const int j = max(min(4 * q + g, n - 1), 0);
          id[q] = xi[j];
          cv[q] = (!ZPAD || 4 * q + g < n) ? tcur[j] : 0.f;
        }
        if constexpr (TSAVE) cl[0] = lane < n ? pf_c : 0.f;
        wave_sync();
      } else {
#pragma unroll
        for (int q = 0; q < NQG; q++) {
          const int j = max(min(4 * q + g, n - 1), 0);
          id[q] = a.row_idx[base + j];
          const float c = a.vals[base + j];
          cv[q] = (!ZPAD || 4 * q + g < n) ? c : 0.f;
        }
        if constexpr (TSAVE) {
#pragma unroll
          for (int s2 = 0; s2 < NSL; s2++) cl[s2] = lane + 64 * s2 < n ? a.vals[base + lane + 64 * s2] : 0.f;
        }
      }
#pragma unroll
      for (int q = 0; q < NQG; q++) {
        const float* src = (!ZPAD || 4 * q + g < n) ? a.X + (size_t)id[q] * k : a.zero_row;
#pragma unroll
        for (int b = 0; b < NV; b++) {
          const int off = b * 16 * VW + i * VW;
          if constexpr (KFULL) {
            const piece_t pc = *reinterpret_cast<const piece_t*>(src + off);
            const float* pf = reinterpret_cast<const float*>(&pc);
#pragma unroll
            for (int c = 0; c < VW; c++) xt[q][b * VW + c] = pf[c];
          } else {
            const piece_t pc = *reinterpret_cast<const piece_t*>(src + min(off, k - VW));
            const float* pf = reinterpret_cast<const float*>(&pc);
#pragma unroll
            for (int c = 0; c < VW; c++) xt[q][b * VW + c] = off < k ? pf[c] : 0.f;
          }
        }
      }
    };
    auto gather = [&](const int base, const int n, const bool from_pf = false) {
      gather_q(std::integral_constant<int, CAPQ>{}, base, n, from_pf);
    };
    // prefix of a streamed row: first sweep -> gather it and park the vectors in LDS; later sweeps -> read them back
    float* pre_lds = sPre + (size_t)wv * PQ * 4 * KP + g * KP;
    auto prefix_chunk = [&](const bool first) {
      if constexpr (PQ > 0) {
        const int base = p1 + tw * PCAP;
        if (first) {
          gather_q(std::integral_constant<int, PQ>{}, base, pw);
#pragma unroll
          for (int q = 0; q < PQ; q++)
#pragma unroll
            for (int b = 0; b < NV; b++) {
              piece_t pc;
              float* pf = reinterpret_cast<float*>(&pc);
#pragma unroll
              for (int c = 0; c < VW; c++) pf[c] = xt[q][b * VW + c];
              *reinterpret_cast<piece_t*>(pre_lds + q * 4 * KP + b * 16 * VW + i * VW) = pc;
            }
        } else {
#pragma unroll
          for (int q = 0; q < PQ; q++) cv[q] = a.vals[base + min(4 * q + g, pw - 1)];
#pragma unroll
          for (int q = 0; q < PQ; q++)
#pragma unroll
            for (int b = 0; b < NV; b++) {
              const piece_t pc = *reinterpret_cast<const piece_t*>(pre_lds + q * 4 * KP + b * 16 * VW + i * VW);
              const float* pf = reinterpret_cast<const float*>(&pc);
#pragma unroll
              for (int c = 0; c < VW; c++) xt[q][b * VW + c] = pf[c];
            }
        }
      }
    };

    float x[RPN], r[RPN], p[RPN], ap[RPN];
    auto load_warm_start = [&]() {
#pragma unroll
      for (int b = 0; b < NV; b++) {
        const int off = b * 16 * VW + i * VW;
        piece_t pc;
        float* pf = reinterpret_cast<float*>(&pc);
#pragma unroll
        for (int c = 0; c < VW; c++) pf[c] = 0.f;
        if (have && off < k) pc = *reinterpret_cast<const piece_t*>(yrow + off);  // warm start
#pragma unroll
        for (int c = 0; c < VW; c++) x[b * VW + c] = pf[c];
      }
    };
    // KFULL: the warm start is requested before the row's vectors (loads return in order), so that the first sweep can
    // run behind the gather quad block by quad block instead of waiting for the warm start = for everything before it
    constexpr bool WS_FIRST = KFULL && resident;
    if constexpr (WS_FIRST) load_warm_start();
    if constexpr (resident) {
      // balanced shares: every wave of the team takes ceil(cnt / WPR) non-zeros rounded up to a quad-pass block (16),
      // so the team's sweep time is that of the average wave, not of a full one next to idle ones
#ifdef RSP_SEQ_FILL
      const int per = CAP;
#else
      const int per = min(CAP, (((cnt + WPR - 1) / WPR) + 15) & ~15);
#endif
      ccnt = max(0, min(per, cnt - tw * per));
#ifdef RSP_GATHER_PRIO   // dev builds: a gathering wave issues ahead of the waves that sweep on the same SIMD
      __builtin_amdgcn_s_setprio(3);
#endif
      if constexpr (WS_FIRST) {
        // no branch around the gather: at a join the compiler has to assume that the warm start is the NEWEST load in
        // flight and drains the queue where it is first used.  A wave without a share gathers the all-zero row (n = 0)
        gather(ccnt > 0 ? p1 + tw * per : p1, ccnt, ccnt > 0 && pf_cnt == ccnt);
      } else {
        if (ccnt > 0) gather(p1 + tw * per, ccnt, pf_cnt == ccnt);
      }
      pf_cnt = -1;
#ifdef RSP_GATHER_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }

    CQ_T(0)   // row switch + gather
    if constexpr (!WS_FIRST) load_warm_start();

    // one pass over the resident quads: t = X_nnz^T v, then acc += X_nnz w  (or the loss terms)
    auto quad_pass = [&](const float(&v)[RPN], const int mode, float(&acc)[RPN], float& lacc, float* tsave) {
      float* trec = (mode == 0 ? tacc : tcur) + g;  // resident rows: slot 4q+g <- t (same value from the 16 lanes)
      constexpr int QB = 4;  // quads per block: 4 independent dot/DPP chains interleave inside one basic block
#pragma unroll
      for (int q0 = 0; q0 < CAPQ; q0 += QB) {
        if (4 * q0 < ccnt) {  // wave-uniform
          float t[QB];
#ifndef RSP_NO_QUAD_ILV
          // the four quads' dot chains and DPP reductions written INTERLEAVED: a packed FMA / DPP add and the instruction
          // that consumes its result need one / two wait states, which four independent chains fill with work
          // (tools/dbg/loop_mix.py counted 21 hazard s_nops per 79-instruction quad block in the chain-after-chain order)
          f32x2 s2[QB];
#pragma unroll
          for (int u = 0; u < QB; u++) s2[u] = f32x2{0.f, 0.f};
#pragma unroll
          for (int rr = 0; rr < RPN; rr += 2) {
            const f32x2 va = {v[rr], v[rr + 1]};
#pragma unroll
            for (int u = 0; u < QB; u++) {
              const f32x2 xa = {xt[q0 + u][rr], xt[q0 + u][rr + 1]};
              s2[u] = __builtin_elementwise_fma(xa, va, s2[u]);
            }
          }
#pragma unroll
          for (int u = 0; u < QB; u++) t[u] = s2[u].x + s2[u].y;
#pragma unroll
          for (int u = 0; u < QB; u++) t[u] += dpp<0xB1>(t[u]);   // quad_perm:[1,0,3,2]
#pragma unroll
          for (int u = 0; u < QB; u++) t[u] += dpp<0x4E>(t[u]);   // quad_perm:[2,3,0,1]
#pragma unroll
          for (int u = 0; u < QB; u++) t[u] += dpp<0x141>(t[u]);  // row_half_mirror
#pragma unroll
          for (int u = 0; u < QB; u++) t[u] += dpp<0x140>(t[u]);  // row_mirror
#else
#pragma unroll
          for (int u = 0; u < QB; u++) {
            // two interleaved partial sums -> v_pk_fma_f32 (RPN is even for every supported rank)
            f32x2 s2 = {0.f, 0.f};
#pragma unroll
            for (int rr = 0; rr < RPN; rr += 2) {
              const f32x2 xa = {xt[q0 + u][rr], xt[q0 + u][rr + 1]};
              const f32x2 va = {v[rr], v[rr + 1]};
              s2 = __builtin_elementwise_fma(xa, va, s2);
            }
            t[u] = s2.x + s2.y;
          }
#pragma unroll
          for (int u = 0; u < QB; u++) t[u] = row16_sum(t[u]);
#endif
#pragma unroll
          for (int u = 0; u < QB; u++) {
            const int q = q0 + u;
            const bool valid = ZPAD || 4 * q + g < ccnt;
            const float c = cv[q];
            if constexpr (STREAM == 1) {  // keep t_j = x_j . v of this sweep: the loss is rebuilt from them
              if (tsave && valid && i == 0) tsave[4 * q + g] = t[u];
            } else if constexpr (TSAVE) {
              trec[4 * q] = t[u];
            }
            if (mode == 2) {
              const float d = IMPLICIT ? ltgt - t[u] : c - t[u];
              lacc += valid ? (IMPLICIT ? c * d * d : d * d) : 0.f;
            } else {
              float w;
              if (mode == 0) w = IMPLICIT ? c - (c - 1.f) * (GB ? t[u] + gbias : t[u]) : c - t[u];
              else w = IMPLICIT ? (c - 1.f) * t[u] : t[u];
              w = valid ? w : 0.f;
#pragma unroll
              for (int rr = 0; rr < RPN; rr++) acc[rr] = fmaf(w, xt[q][rr], acc[rr]);
            }
          }
        }
      }
    };

    // mode 0: out = X_nnz (c - c1 % X_nnz^T v) - G v ; mode 1: out = X_nnz (c1 % X_nnz^T v) + G v ; mode 2: loss
    auto sweep = [&](const float(&v)[RPN], const int mode, float(&out)[RPN], float& loss_out, const bool live,
                     const int sidx = 0) {
      float acc[RPN];
#pragma unroll
      for (int rr = 0; rr < RPN; rr++) acc[rr] = 0.f;
      float lacc = 0.f;
      float vinv = 0.f;   // DMF: 2^-(scale of v) * 2^-(scale of G) with the sign of the mode
      if constexpr (DMF) {
        if (mode != 2) {
          // (1) publish v as two fp16 terms of v * 2^ve (every wave, live or not: the four columns are independent)
          float vmax = 0.f;
#pragma unroll
          for (int rr = 0; rr < RPN; rr++) vmax = fmaxf(vmax, fabsf(v[rr]));
          vmax = row16_max(vmax);
          const int ve = fp16_scale_exp(vmax);
          const float vs = __uint_as_float((unsigned)ve << 23);
          vinv = __uint_as_float((unsigned)(254 - ve) << 23) * (mode == 0 ? -1.f : 1.f);
          if (g < NV) {   // group b publishes piece b: elements [64 b + 4 i, + 4)
            float w4[VW];
#pragma unroll
            for (int c = 0; c < VW; c++) w4[c] = (g == 0 ? v[c] : v[VW + c]) * vs;
            unsigned h0, l0, h1, l1;
            split_f16(w4[0], w4[1], h0, l0);
            split_f16(w4[2], w4[3], h1, l1);
            const int off = wv * SM::dmf_ps + g * 16 * VW + i * VW;
            *reinterpret_cast<uint2*>(sPh + off) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(sPl + off) = make_uint2(l0, l1);
          }
          __syncthreads();
          // (2) this wave's 32 rows of G against the four vectors (columns 4..15 of the tile repeat them)
          f32x4 d0[2], d1[2];
#pragma unroll
          for (int t = 0; t < 2; t++) d0[t] = d1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
          const int nb = (lane & 3) * SM::dmf_ps + 8 * (lane >> 4);
#pragma unroll
          for (int ks = 0; ks < 4; ks++) {
            const f16x8 bh = *reinterpret_cast<const f16x8*>(sPh + nb + 32 * ks);
            const f16x8 bl = *reinterpret_cast<const f16x8*>(sPl + nb + 32 * ks);
#pragma unroll
            for (int t = 0; t < 2; t++) {
              f16x8 ah, al;
              if constexpr (DMF == 1) {
                ah = gAh[t][ks];
                al = gAl[t][ks];
              } else {
                ah = *reinterpret_cast<const f16x8*>(sGf + ((t * 4 + ks) * 2 + 0) * 1024 + lane * 16);
                al = *reinterpret_cast<const f16x8*>(sGf + ((t * 4 + ks) * 2 + 1) * 1024 + lane * 16);
              }
              d0[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, d0[t], 0, 0, 0);
              d1[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, d1[t], 0, 0, 0);
              d1[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, d1[t], 0, 0, 0);
            }
          }
          if ((lane & 15) < 4) {   // D: column = lane & 15, rows 4 (lane >> 4) + 0..3 of the tile
#pragma unroll
            for (int t = 0; t < 2; t++) {
              const f32x4 d = (d0[t] + d1[t]) * ginv;
              *reinterpret_cast<f32x4*>(sOut + (lane & 15) * SM::dmf_os + 32 * wv + 16 * t + 4 * (lane >> 4)) = d;
            }
          }
        }
      }
      if (live) {
        if constexpr (IMPLICIT && !DMF) {
          if (mode != 2) {  // publish v for the broadcast reads of the G*v product
            wave_sync();
            if (g == 0) {
#pragma unroll
              for (int b = 0; b < NV; b++) {
                piece_t pc;
                float* pf = reinterpret_cast<float*>(&pc);
#pragma unroll
                for (int c = 0; c < VW; c++) pf[c] = v[b * VW + c];
                *reinterpret_cast<piece_t*>(vec + b * 16 * VW + i * VW) = pc;
              }
            }
            wave_sync();
          }
        }
        // dense part: acc -/+= G v, the groups split the rows of G.  Resident rows run it before the quad pass (its
        // LDS reads are in flight meanwhile); streamed rows after, where acc is not live across the gathers.
        auto dense_part = [&]() {
          if (IMPLICIT && !DMF && mode != 2) {
            const float sign = mode == 0 ? -1.f : 1.f;
            if constexpr (NRES > 0) {
#pragma unroll
              for (int rs = 0; rs < NRES; rs++) {
                const int kk = 4 * (g + 4 * rs);
                const float4 vb = *reinterpret_cast<const float4*>(vec + kk);
                const float vv[4] = {sign * vb.x, sign * vb.y, sign * vb.z, sign * vb.w};
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                  for (int b = 0; b < NV; b++) {
                    const float* pf = reinterpret_cast<const float*>(&gres[rs][u][b]);
#pragma unroll
                    for (int c = 0; c < VW; c++) acc[b * VW + c] = fmaf(vv[u], pf[c], acc[b * VW + c]);
                  }
              }
            }
            for (int s4 = tw * 4 + g + 4 * NRES; s4 < KP / 4; s4 += WPR * 4) {
              const int kk = 4 * s4;
              const float4 vb = *reinterpret_cast<const float4*>(vec + kk);
              const float vv[4] = {sign * vb.x, sign * vb.y, sign * vb.z, sign * vb.w};
#pragma unroll
              for (int u = 0; u < 4; u++) {
                const float* grow = sG + (kk + u) * KP;
#pragma unroll
                for (int b = 0; b < NV; b++) {
                  const piece_t pc = *reinterpret_cast<const piece_t*>(grow + b * 16 * VW + i * VW);
                  const float* pf = reinterpret_cast<const float*>(&pc);
#pragma unroll
                  for (int c = 0; c < VW; c++) acc[b * VW + c] = fmaf(vv[u], pf[c], acc[b * VW + c]);
                }
              }
            }
          }
        };
        if constexpr (GVFIRST) dense_part();
        if (resident) {
          if (mode != 2 || !TSAVE) {
            quad_pass(v, mode, acc, lacc, nullptr);
          } else {
            // loss from t_acc = X_nnz^T y built up by the sweeps (the vectors are not touched again)
            wave_sync();
            float esum = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < NSL; s2++) {
              const int sl = NSL == 1 ? (lane & (CAP - 1)) : lane + 64 * s2;
              const float t = tacc[sl];
              const float clv = cl[s2];
              const float d = IMPLICIT ? ltgt - t : clv - t;
              const float e = IMPLICIT ? clv * d * d : d * d;
              esum += lane + 64 * s2 < ccnt ? e : 0.f;
            }
            lacc = row16_sum(esum);  // groups_sum below finishes the wave sum
          }
        } else if (mode == 2 && tscr) {
          // streamed rows: t_final = t_0 + sum_s alpha_s t_s from the scratch written by the sweeps -- 20 bytes
          // per non-zero instead of re-gathering its 512-byte vector
          for (int ch = tw; ch < (cnt + CAP - 1) / CAP; ch += WPR) {   // all positions of the row, prefix included
            const int n = min(CAP, cnt - ch * CAP);
            if (lane < n) {
              const size_t pos = (size_t)ch * CAP + lane;
              float t = tscr[pos];
#pragma unroll
              for (int s2 = 0; s2 < kMaxSavedSweeps; s2++)
                if (s2 < a.cg_steps) t = fmaf(alph[s2], tscr[(size_t)(s2 + 1) * a.stream_nnz + pos], t);
              const float c = a.vals[p1 + pos];
              const float d = IMPLICIT ? ltgt - t : c - t;
              lacc += IMPLICIT ? c * d * d : d * d;
            }
          }
          lacc = row16_sum(lacc);  // groups_sum below finishes the wave sum
        } else {
          if (pw > 0) {  // wave-uniform
            prefix_chunk(mode == 0);   // mode 0 is the first sweep of every row
            ccnt = pw;
            quad_pass(v, mode, acc, lacc, tscr ? tscr + (size_t)sidx * a.stream_nnz + (size_t)tw * PCAP : nullptr);
          }
          for (int ch = tw; ch < nchunks; ch += WPR) {
            ccnt = min(CAP, cnt - pre - ch * CAP);
            gather(p1 + pre + ch * CAP, ccnt, IDXPF && pf_pos == ch && pf_cnt == ccnt);
            if constexpr (IDXPF) {  // this wave's next chunk: further down the row, or the first one of the next sweep
              const int nch = ch + WPR < nchunks ? ch + WPR : tw;
              const int ncc = min(CAP, cnt - pre - nch * CAP);
              const int j = p1 + pre + nch * CAP + min(lane & (CAP - 1), ncc - 1);
              pf_id = a.row_idx[j];
              pf_c = a.vals[j];
              pf_cnt = ncc;
              pf_pos = nch;
            }
            quad_pass(v, mode, acc, lacc,
                      tscr ? tscr + (size_t)sidx * a.stream_nnz + (size_t)pre + (size_t)ch * CAP : nullptr);
          }
        }
        if constexpr (!GVFIRST) dense_part();
        if constexpr (!DMF) {
          if (mode != 2) {
            if constexpr (GRS) {
              if constexpr (WPR == 1) groups_all_reduce8(acc);   // (teams: scattered where the partials are published)
            } else {
#pragma unroll
              for (int rr = 0; rr < RPN; rr++) acc[rr] = groups_sum(acc[rr]);
            }
          } else {
            lacc = groups_sum(lacc);
          }
        }
      }
      if constexpr (DMF) {
        if (mode != 2) {
          // (3) the products are complete: fold this row's G v into group 0's partial sums, then all-reduce the groups
          __syncthreads();
          const float f = g == 0 ? vinv : 0.f;
#pragma unroll
          for (int b = 0; b < NV; b++) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(sOut + wv * SM::dmf_os + b * 16 * VW + i * VW);
#pragma unroll
            for (int c = 0; c < VW; c++) acc[b * VW + c] = fmaf(f, o[c], acc[b * VW + c]);
          }
          if constexpr (GRS) {
            groups_all_reduce8(acc);
          } else {
#pragma unroll
            for (int rr = 0; rr < RPN; rr++) acc[rr] = groups_sum(acc[rr]);
          }
        } else {
          lacc = groups_sum(lacc);
        }
      }
      if constexpr (WPR > 1) {
        float* red = sRed + buf * WAVES * KP;
        if (mode != 2) {
          if constexpr (GRS) {
            // the groups' partial sums meet on the way to LDS: row g of r0 / r1 = the wave's total of register g / 4 + g, i.e.
            // of the vector elements 4 i + g and 64 + 4 i + g -- every lane publishes two floats (a permutation of the 128)
            const float r0 = groups_reduce_scatter4(acc[0], acc[1], acc[2], acc[3]);
            const float r1 = groups_reduce_scatter4(acc[4], acc[5], acc[6], acc[7]);
            red[wv * KP + 4 * i + g] = r0;
            red[wv * KP + 64 + 4 * i + g] = r1;
          } else if (g == 0) {
#pragma unroll
            for (int b = 0; b < NV; b++) {
              piece_t pc;
              float* pf = reinterpret_cast<float*>(&pc);
#pragma unroll
              for (int c = 0; c < VW; c++) pf[c] = acc[b * VW + c];
              *reinterpret_cast<piece_t*>(red + wv * KP + b * 16 * VW + i * VW) = pc;
            }
          }
        } else if (lane == 0) {
          sRedL[buf * WAVES + wv] = lacc;
        }
        __syncthreads();
        if (mode != 2) {
#pragma unroll
          for (int rr = 0; rr < RPN; rr++) acc[rr] = 0.f;
#pragma unroll
          for (int w2 = 0; w2 < WPR; w2++) {
#pragma unroll
            for (int b = 0; b < NV; b++) {
              const piece_t pc = *reinterpret_cast<const piece_t*>(red + (team * WPR + w2) * KP + b * 16 * VW + i * VW);
              const float* pf = reinterpret_cast<const float*>(&pc);
#pragma unroll
              for (int c = 0; c < VW; c++) acc[b * VW + c] += pf[c];
            }
          }
        } else {
          lacc = 0.f;
#pragma unroll
          for (int w2 = 0; w2 < WPR; w2++) lacc += sRedL[buf * WAVES + team * WPR + w2];
        }
        buf ^= 1;
      } else {
        // no barrier on this path: pin the schedule at sweep boundaries like the barrier does for teams,
        // otherwise the scheduler hoists the next sweep's LDS/global loads and the allocator spills
        __builtin_amdgcn_sched_barrier(0);
      }
      if (mode != 2) {
#pragma unroll
        for (int rr = 0; rr < RPN; rr++) out[rr] = acc[rr];
        if constexpr (GB) {
          if (mode == 0) {   // + global_bias_base (the same k floats for every row: L2)
#pragma unroll
            for (int b = 0; b < NV; b++) {
              const int off = b * 16 * VW + i * VW;
              const piece_t pc = *reinterpret_cast<const piece_t*>(a.rhs_init + min(off, k - VW));
              const float* pf = reinterpret_cast<const float*>(&pc);
#pragma unroll
              for (int c = 0; c < VW; c++) out[b * VW + c] += off < k ? pf[c] : 0.f;
            }
          }
        }
        if constexpr (!IMPLICIT) {
#pragma unroll
          for (int rr = 0; rr < RPN; rr++) out[rr] = fmaf(mode == 0 ? -lam_use : lam_use, v[rr], out[rr]);
        }
      } else {
        loss_out = lacc;
      }
    };

    auto dot16 = [&](const float(&u)[RPN], const float(&w)[RPN]) {
      float s = 0.f;
#pragma unroll
      for (int rr = 0; rr < RPN; rr++) s = fmaf(u[rr], w[rr], s);
      return row16_sum(s);  // the 16 lanes of a group cover the whole vector; groups are replicas
    };

    const bool live = have && (GB || cnt > 0);
    float dummy = 0.f;
    CQ_T(1)   // warm start, setup
    sweep(x, 0, r, dummy, live);
    CQ_T(2)   // sweeps
    if constexpr (IDXPF && STREAM == 0) {
      // the next row's pointers were requested at the top of this iteration and have arrived by now
      if (it + 1 < rows_per_team && row_index(it + 1) < n_rows) {
        const int np1 = rfl(p1_c), ncnt = rfl(p2_c) - np1;
        const int nper = min(CAP, (((ncnt + WPR - 1) / WPR) + 15) & ~15);
        const int ncc = max(0, min(nper, ncnt - tw * nper));
        if (ncc > 0) {
          const int j = np1 + tw * nper + min(lane & (CAP - 1), ncc - 1);
          pf_id = a.row_idx[j];
          pf_c = a.vals[j];
          pf_cnt = ncc;
        }
      }
    }
    if constexpr (DMAPF) {
      if (it + 1 < rows_per_team && row_index(it + 1) < n_rows) {
        const int np1 = rfl(p1_c), ncnt = rfl(p2_c) - np1;
        const int nper = min(CAP, (((ncnt + WPR - 1) / WPR) + 15) & ~15);
        const int ncc = max(0, min(nper, ncnt - tw * nper));
        if (ncc > 0) {   // (the slot's previous content was consumed by this row's gather)
          const int j = np1 + tw * nper + min(lane, ncc - 1);
          dma4(a.row_idx + j, lds_addr(pfxI));
          dma4(a.vals + j, lds_addr(pfxC));
          pf_cnt = ncc;
        }
      }
    }
    CQ_T(3)   // staging issue / index prefetch
#pragma unroll
    for (int rr = 0; rr < RPN; rr++) p[rr] = r[rr];
    float rsold = dot16(r, r);
    bool conv = false;
    for (int itc = 0; itc < a.cg_steps; ++itc) {
      if (WPR == 1 && !DMF && conv) break;
      CQ_T(4)   // CG scalars and updates
      sweep(p, 1, ap, dummy, live && !conv, itc + 1);
      CQ_T(2)
      CQ_T(3)
      if (!conv) {
        const float pap = dot16(p, ap);
        // rsold / alpha / beta as the reference holds them: double scalars fed by T-valued dot products
        // (wrmf_implicit.hpp:18-27), rounded to float where they meet the vectors
        const float alpha = (float)((double)rsold / (double)pap);
#pragma unroll
        for (int s2 = 0; s2 < kMaxSavedSweeps; s2++)
          if (s2 == itc) alph[s2] = alpha;
        if constexpr (TSAVE) {  // t_acc += alpha * t_cur  (x += alpha p  =>  X_nnz^T x += alpha X_nnz^T p)
          wave_sync();
#pragma unroll
          for (int s2 = 0; s2 < NSL; s2++) {
            const int sl = NSL == 1 ? (lane & (CAP - 1)) : lane + 64 * s2;
            tacc[sl] = fmaf(alpha, tcur[sl], tacc[sl]);
          }
          wave_sync();
        }
#pragma unroll
        for (int rr = 0; rr < RPN; rr++) {
          x[rr] = fmaf(alpha, p[rr], x[rr]);
          r[rr] = fmaf(-alpha, ap[rr], r[rr]);
        }
        const float rsnew = dot16(r, r);
        if (rsnew < kCgTolQ) {
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
    CQ_T(4)
    sweep(x, 2, ap, rl, live);
    CQ_T(2)
    if constexpr (DMAPF) wait_vm0();   // the announced row's indices have long landed; nothing of this wave's is in flight after this
    if constexpr (IDXPF && STREAM == 0) {
      // settle the prefetch registers HERE (their loads were issued three sweeps ago): left to the next row's first use, the
      // wait sits behind that row's warm-start request -- loads return in order -- and the vectors are requested a round trip late
      asm volatile("" : "+v"(pf_id), "+v"(pf_c));
    }
    if (live && tw == 0) {
      const float xx = dot16(x, x);
      wloss += IMPLICIT ? (double)rl + a.lambda_loss * (double)xx : (double)(rl + lam_use * xx);
      if (g == 0) {
#pragma unroll
        for (int b = 0; b < NV; b++) {
          const int off = b * 16 * VW + i * VW;
          if (off < k) {
            piece_t pc;
            float* pf = reinterpret_cast<float*>(&pc);
#pragma unroll
            for (int c = 0; c < VW; c++) pf[c] = x[b * VW + c];
            *reinterpret_cast<piece_t*>(yrow + off) = pc;
          }
        }
      }
    }
  }
  CQ_T(5)   // loss, store of the row
#ifdef RSP_CGQ_PROF
  if (WAVES == 8 && WPR == 8 && !STREAM && a.ne_prof && lane == 0)
    for (int j = 0; j < 6; j++) atomicAdd(a.ne_prof + (size_t)65536 * 80 - 16 + j, cq_t[j]);
#endif
  if (lane == 0) a.loss_partials[loss_slot0 + (size_t)blockIdx.x * WAVES + wv] = wloss;
}

// Launch table.  Rows are bucketed by length; each bucket is one launch of the kernel instantiated for
// (waves per workgroup W, waves per row WPR, resident quads per wave, streamed?):
//   bucket 0  streamed rows (longer than 512 non-zeros): teams of 8 waves, 512-thread workgroups
//   bucket 1  257..512 non-zeros: resident on 8-wave teams
//   bucket 2.. 129..256 / 65..128 / 33..64 non-zeros: resident on teams of 4 / 2 / 1 waves of 256-thread
//             workgroups (two per CU: they run out of phase, one gathers while the other sweeps)
//   bucket 5  <= 32 non-zeros: one wave per row with a 32-slot tile (half the redundant gather slots)
// Every instantiation is capped at 256 VGPRs (2 waves per SIMD, 8 waves per CU).
struct BucketDef { int waves, wpr, capq, stream, max_len; };
constexpr int kNB = 6;
// Geometry 1 (rank 33..64, round 3): a rank-64 vector is 4 registers of a lane, so the 128 registers that hold 64 vectors at
// rank 128 hold 128 -- the same row-length classes run on HALF the waves: 257..512 on 4-wave teams of 256-thread
// workgroups (two per CU: gather and sweeps of the two overlap, which the one 512-thread workgroup per CU of geometry 0
// cannot), 129..256 on 2 waves, 65..128 on one.  The class boundaries are those of geometry 0: the schedule is built once
// per matrix, before the rank is known.
constexpr int kNCfg = 2;
constexpr BucketDef kBuckets[kNCfg][kNB] = {
    {{8, 8, 16, 1, 0x7fffffff}, {8, 8, 16, 0, 512}, {4, 4, 16, 0, 256}, {4, 2, 16, 0, 128}, {4, 1, 16, 0, 64}, {4, 1, 8, 0, 32}},
    {{8, 8, 16, 1, 0x7fffffff}, {4, 4, 32, 0, 512}, {4, 2, 32, 0, 256}, {4, 1, 32, 0, 128}, {4, 1, 16, 0, 64}, {4, 1, 8, 0, 32}},
};
constexpr int cfg_of_kp(int KP) { return KP == 64 ? 1 : 0; }

// dev builds (-DRSP_AB): RSPARSE_HIP_DENSE_MFMA=0 keeps the one-wave rows of <= 32 non-zeros on the vector-unit G v
// product (the A/B switch behind DESIGN.md 3.1)
bool dense_mfma_enabled() {
#ifdef RSP_AB
  static const bool on = [] {
    const char* e = std::getenv("RSPARSE_HIP_DENSE_MFMA");
    return !(e && e[0] == '0');
  }();
  return on;
#else
  return true;
#endif
}

// dev builds (-DRSP_AB): RSPARSE_HIP_KFULL=0 keeps the instantiations that drain the gather before the first sweep
bool kfull_enabled() {
#ifdef RSP_AB
  static const bool on = [] {
    const char* e = std::getenv("RSPARSE_HIP_KFULL");
    return !(e && e[0] == '0');
  }();
  return on;
#else
  return true;
#endif
}

template <int KP, int WAVES, int CAPQ, int WPR, int STREAM, bool IMPLICIT, bool GB, int DMF = 0, bool KFULL = false>
hipError_t launch_bucket(const AlsArgs& a, const int32_t* rows, int n_rows, int grid, size_t slot0, hipStream_t s,
                         hipEvent_t* ev_slot) {
  if (n_rows <= 0) return hipSuccess;
  constexpr bool kDmfGeometry = IMPLICIT && KP == 128 && WAVES == 4 && WPR == 1 && (CAPQ == 8 || CAPQ == 16) && STREAM == 0;
  if constexpr (kDmfGeometry && !DMF) {
    if (dense_mfma_enabled())
      return launch_bucket<KP, WAVES, CAPQ, WPR, STREAM, IMPLICIT, GB, CAPQ == 8 ? 1 : 2>(a, rows, n_rows, grid, slot0, s, ev_slot);
  }
  // the rank is the padded rank (32 / 64 / 128): the instantiation whose first sweep runs behind the gather
  if constexpr (!KFULL && STREAM == 0) {
    if (a.k == KP && kfull_enabled())
      return launch_bucket<KP, WAVES, CAPQ, WPR, STREAM, IMPLICIT, GB, DMF, true>(a, rows, n_rows, grid, slot0, s, ev_slot);
  }
  constexpr int TEAMS = WAVES / WPR;
  auto kern = als_cgq_kernel<KP, CAPQ, WAVES, WPR, STREAM, IMPLICIT, DMF, GB, KFULL>;
  const size_t lds = QSmem<KP, CAPQ, WAVES, WPR, STREAM, IMPLICIT, DMF>::bytes;
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return err;
  const int total_teams = grid * TEAMS;
  const int rpt = (n_rows + total_teams - 1) / total_teams;
  prof_note(ev_slot, reinterpret_cast<const void*>(kern));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, s, a, rows, n_rows, rpt, slot0);
  return hipGetLastError();
}

// Side streams for overlapping the bucket launches (they touch disjoint rows): forked from / joined to the
// caller's stream with events, so the call stays asynchronous and ordered on that stream.
struct BucketStreams {
  hipStream_t st[kNB] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t fork = nullptr, fork0 = nullptr;   // fork0: in front of the first bucket's two launches (round 6)
  hipEvent_t done[kNB] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int device = -1;
  hipError_t ensure() {
    int dev = 0;
    hipError_t err = hipGetDevice(&dev);
    if (err != hipSuccess) return err;
    if (dev == device) return hipSuccess;
    device = dev;  // streams belong to a device; a process drives one GPU, so this happens once
    if ((err = hipEventCreateWithFlags(&fork, hipEventDisableTiming)) != hipSuccess) return err;
    if ((err = hipEventCreateWithFlags(&fork0, hipEventDisableTiming)) != hipSuccess) return err;
    for (int b = 0; b < kNB; b++) {
      if ((err = hipStreamCreateWithFlags(&st[b], hipStreamNonBlocking)) != hipSuccess) return err;
      if ((err = hipEventCreateWithFlags(&done[b], hipEventDisableTiming)) != hipSuccess) return err;
    }
    return hipSuccess;
  }
};
thread_local BucketStreams g_bs;

// 0: every launch on the caller's stream; 1: every bucket on its own side stream; 2: the long-row launch first on the
// caller's stream (it occupies every CU's LDS: nothing runs beside it), the resident buckets on side streams after it
// (rsparse_hip_set_launch_mode; measured on the bench line: 0 -> 165.5 ms, 2 -> 166.0 ms, 1 -> 171.8 ms per iteration)
int g_launch_mode = 2;
int concurrent_buckets() { return g_launch_mode; }

// loss slots of bucket b: one per list entry of the normal-equation launch, else one per wave of the bucket's launch(es)
size_t bucket_slots(const QSchedule& q, int b, int k, bool implicit) {
  const int cfg = cfg_of_kp(padded_rank(k));
  const BucketDef d = kBuckets[cfg][b];
  if (d.wpr <= 0) return 0;
  if (d.stream && ne_supported(k)) return (size_t)(q.ne_entries + q.ne_nsplit) + (size_t)(q.mf_n > 0 ? cg_mf_loss_slots(q.mf_n) : 0);
  const int rows = q.off[b + 1] - q.off[b];
  if (b == kNB - 1 && rows > 0 && cgp_supported(k, implicit) && dense_mfma_enabled()) {
    const int split = std::min(std::max(q.pair_first, q.off[b]), q.off[b + 1]);
    return (size_t)cgq_bucket_grid(split - q.off[b], b, cfg) * d.waves + (size_t)cgp_grid(q.off[b + 1] - split) * 4;
  }
  return (size_t)cgq_bucket_grid(rows, b, cfg) * d.waves;
}

template <int KP, int CFG, bool IMPLICIT, bool GB>
hipError_t launch_all(const AlsArgs& a, const QSchedule& q, hipStream_t s, hipEvent_t* ev) {
  hipError_t err;
  size_t slot = 0;
  // per-kernel timing (ev != nullptr) needs the launches back to back on one stream
  const int cmode = ev ? 0 : concurrent_buckets();
  const bool overlap = cmode != 0;
  bool forked = false;
  if (overlap) {
    if ((err = g_bs.ensure()) != hipSuccess) return err;
  }
#define RSP_BUCKET(B)                                                                                       \
  {                                                                                                         \
    constexpr BucketDef D = kBuckets[CFG][B];                                                               \
    if (ev && (err = hipEventRecord(ev[B], s)) != hipSuccess) return err;                                   \
    if constexpr (D.wpr > 0) {                                                                              \
      const int n = q.off[B + 1] - q.off[B];                                                                \
      const int grid = cgq_bucket_grid(n, B, CFG);                                                          \
      if (n > 0) {                                                                                          \
        hipStream_t bs = s;                                                                                 \
        const bool side = overlap && !(cmode == 2 && D.stream);                                             \
        if (side) {                                                                                         \
          if (!forked) {                                                                                    \
            if ((err = hipEventRecord(g_bs.fork, s)) != hipSuccess) return err;                             \
            forked = true;                                                                                  \
          }                                                                                                 \
          bs = g_bs.st[B];                                                                                  \
          if ((err = hipStreamWaitEvent(bs, g_bs.fork, 0)) != hipSuccess) return err;                       \
        }                                                                                                   \
        if (D.stream && ne_supported(a.k)) {                                                                \
          /* rank 128 (round 6): the rows up to kCgMfMax non-zeros one wave per row (wrmf_cg_mf.hip) on the bucket's side stream, */ \
          /* the giant rows on the normal-equation kernel beside it (few workgroups: they start first, they end last) */ \
          /* (the giant rows' launch is SUBMITTED first: the wave-per-row workgroups live as long as their launch and */ \
          /*  leave no register file for a partner, so a launch submitted behind them starts when they end) */ \
          const bool mside = overlap && q.ne_wg > 0 && q.mf_n > 0;                                          \
          /* (a fork point of its own in FRONT of the giant rows' launch; the other buckets still fork behind both) */ \
          if (mside && (err = hipEventRecord(g_bs.fork0, s)) != hipSuccess) return err;                     \
          if (q.ne_wg > 0 &&                                                                                \
              (err = launch_als_ne(a, q, IMPLICIT, a.loss_partials + slot, bs, q.mf_n > 0 ? nullptr : (ev ? ev + B : nullptr))) != hipSuccess) \
            return err;                                                                                     \
          if (q.mf_n > 0) {                                                                                 \
            hipStream_t ms = bs;                                                                            \
            if (mside) {                                                                                    \
              ms = g_bs.st[B];                                                                              \
              if ((err = hipStreamWaitEvent(ms, g_bs.fork0, 0)) != hipSuccess) return err;                  \
            }                                                                                               \
            if ((err = launch_als_cg_mf(a, q.mf_rows, q.mf_n, (int)(slot + (size_t)(q.ne_entries + q.ne_nsplit)), ms, \
                                        ev ? ev + B : nullptr)) != hipSuccess)                              \
              return err;                                                                                   \
            if (mside) {                                                                                    \
              if ((err = hipEventRecord(g_bs.done[B], ms)) != hipSuccess) return err;                       \
              if ((err = hipStreamWaitEvent(s, g_bs.done[B], 0)) != hipSuccess) return err;                 \
            }                                                                                               \
          }                                                                                                 \
        } else if constexpr (D.stream && KP > 32) {   /* ranks above 32 always take the branch above */     \
          return hipErrorInvalidValue;                                                                      \
        } else if (B == kNB - 1 && cgp_supported(a.k, IMPLICIT) && dense_mfma_enabled()) {                    \
          /* the last bucket in two launches: rows of 17..32 non-zeros one per wave, the rest two per wave (wrmf_cgp.hip) */ \
          const int first = q.off[B], split = std::min(std::max(q.pair_first, first), q.off[B + 1]);         \
          const int n_main = split - first, n_pair = q.off[B + 1] - split;                                   \
          const int g_main = cgq_bucket_grid(n_main, B, CFG);                                                \
          if ((err = launch_bucket<KP, D.waves, D.capq, D.wpr, D.stream, IMPLICIT, GB>(a, q.order + first, n_main, g_main, slot, \
                                                                                  bs, ev ? ev + B : nullptr)) != hipSuccess) \
            return err;                                                                                     \
          if ((err = launch_als_cgp(a, q.order + split, n_pair, slot + (size_t)g_main * D.waves, bs,          \
                                    n_main > 0 ? nullptr : (ev ? ev + B : nullptr))) != hipSuccess)           \
            return err;                                                                                     \
        } else if ((err = launch_bucket<KP, D.waves, D.capq, D.wpr, D.stream, IMPLICIT, GB>(a, q.order + q.off[B], n, grid, slot, \
                                                                                bs, ev ? ev + B : nullptr)) != hipSuccess) \
          return err;                                                                                       \
        if (side) {                                                                                         \
          if ((err = hipEventRecord(g_bs.done[B], bs)) != hipSuccess) return err;                           \
          if ((err = hipStreamWaitEvent(s, g_bs.done[B], 0)) != hipSuccess) return err;                     \
        }                                                                                                   \
      }                                                                                                     \
      slot += bucket_slots(q, B, a.k, IMPLICIT);                                                             \
    }                                                                                                       \
  }
  RSP_BUCKET(0)
  RSP_BUCKET(1)
  RSP_BUCKET(2)
  RSP_BUCKET(3)
  RSP_BUCKET(4)
  RSP_BUCKET(5)
#undef RSP_BUCKET
  if (ev && (err = hipEventRecord(ev[kNB], s)) != hipSuccess) return err;
  return hipSuccess;
}

}  // namespace

int cgq_num_buckets() { return kNB; }
int cgq_bucket_wpr(int cfg, int b) { return kBuckets[cfg][b].wpr; }
int cgq_bucket_capq(int cfg, int b) { return kBuckets[cfg][b].capq; }
int cgq_bucket_waves(int cfg, int b) { return kBuckets[cfg][b].waves; }
int cgq_bucket_stream(int cfg, int b) { return kBuckets[cfg][b].stream; }

int cgq_bucket_grid(int n_rows, int b, int cfg) {
  const BucketDef d = kBuckets[cfg][b];
  if (n_rows <= 0 || d.wpr <= 0) return 0;
  const int teams = d.waves / d.wpr;
  // rows per team: amortises the per-workgroup start-up (64 KB Gramian load, LDS clear); the streamed
  // bucket holds few, very long rows and keeps a small quota for balance
  // (measured on config 3: doubling the quota of the team kernels from 4 / 16 is worth 3 % of the iteration, a
  // further doubling is flat; the one-wave kernels prefer 64)
  const int base = d.stream ? 16 : (d.wpr == 1 ? 64 : (d.wpr == 2 ? 64 : 32));
  int rows_per_team = base;
  // small buckets (shards of a multi-GPU run, tiny matrices): spread the rows over the CUs first -- the quota
  // only grows once there are two workgroups per CU
  const long spread = (long)teams * 512;
  if ((long)n_rows < spread * rows_per_team) rows_per_team = (int)((n_rows + spread - 1) / spread);
  if (rows_per_team < 1) rows_per_team = 1;
  const long per_wg = (long)teams * rows_per_team;
  long grid = (n_rows + per_wg - 1) / per_wg;
  if (grid < 1) grid = 1;
  return (int)grid;
}

int cgq_bucket_of(int len, int cfg) {  // last (smallest-team) bucket whose capacity holds the row
  int best = 0;
  for (int b = 0; b < kNB; b++)
    if (kBuckets[cfg][b].wpr > 0 && len <= kBuckets[cfg][b].max_len) best = b;
  return best;
}

size_t cgq_loss_slots(const QSchedule& q, int k, bool implicit) {
  size_t n = 0;
  for (int b = 0; b < kNB; b++) n += bucket_slots(q, b, k, implicit);
  return n;
}

int cgq_default_cfg() { return 0; }
void cgq_set_launch_mode(int mode) { g_launch_mode = mode; }

hipError_t launch_als_cgq(const AlsArgs& a, const QSchedule& q, bool implicit, hipStream_t s, hipEvent_t* ev) {
  const int KP = padded_rank(a.k);
#define RSP_DISPATCH(KPV)                                                                                   \
  if (KP == KPV) {                                                                                          \
    if (implicit && a.gbias != 0.f) return launch_all<KPV, cfg_of_kp(KPV), true, true>(a, q, s, ev);        \
    return implicit ? launch_all<KPV, cfg_of_kp(KPV), true, false>(a, q, s, ev)                             \
                    : launch_all<KPV, cfg_of_kp(KPV), false, false>(a, q, s, ev);                           \
  }
  RSP_DISPATCH(32)
  RSP_DISPATCH(64)
  RSP_DISPATCH(128)
#undef RSP_DISPATCH
  return hipErrorInvalidValue;
}

}  // namespace rsparse_hip
