// Long rows: ONE pass over the row's non-zeros, normal equations on the matrix cores (gfx950, wave64).
//
// The register-resident CG kernels (wrmf_cgq.hip) touch every gathered factor vector 2 (cg_steps + 1) times; that is
// free while the row fits the CU's register file, but rows beyond 512 non-zeros had to be re-gathered in every CG
// sweep (3.7x the algorithmic HBM traffic, round 1).  Here such a row is gathered exactly once:
//
//     M1 = X_nnz diag(c - 1) X_nnz^T   (implicit)            M2 = X_nnz X_nnz^T
//     b  = X_nnz c                     (rhs, wrmf_implicit.hpp:207-208 / wrmf_explicit.hpp:99-101)
//
// are accumulated on the MFMA pipes while the vectors stream through, and the per-row solve then runs on the k x k
// system  A = XtX + M1  (explicit: A = M2 + lambda_use I)  held in LDS: cg_solver_implicit / cg_solver_explicit
// (wrmf_implicit.hpp:8-32, wrmf_explicit.hpp:8-31) with  A p  in place of  XtX p + X_nnz ((c-1) % X_nnz^T p)  -- the
// same operator, evaluated from the assembled matrix.  The loss  sum_j c_j (1 - x_j.y)^2  (wrmf_implicit.hpp:259-261)
// needs no second pass either: it equals  sum c - 2 y.b + y^T (M1 + M2) y, and the quadratic forms are read off the
// accumulators.
//
// Arithmetic: fp32 MFMA runs at the vector rate (157 TF), bf16 MFMA 16x faster.  Every fp32 operand is split exactly
// into NS bf16 terms (x = x1 + x2 [+ x3], residuals are exact in fp32) and the products of total order < NS are
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16: NS = 3 keeps 6 products and is accurate to ~2^-23 per product, i.e.
// fp32-equivalent; M2 only feeds the loss and always uses the 3 products of its first two terms (2^-16).
//
// Structure: a workgroup = 4 waves = one row at a time, rows from a per-workgroup list balanced on the host (longest
// processing time first).  The row's 16-non-zero steps go round robin to the workgroup's RING GROUPS: a group is the
// set of waves that consume the same steps (one wave, or a PAIR of waves that split the accumulator tiles when one wave
// cannot hold them all).  Each group owns a ring of D slots in LDS that it fills by LDS-DMA
// (global_load_lds_dwordx4: whole 512-byte vectors, no staging registers; every wave of the group issues its share,
// because one wave keeps only ~10 KB of LDS-DMA in flight) D-1 steps ahead; the stream of steps runs
// ACROSS row boundaries, so the gather of the next row is in flight while the current row is solved.  MFMA operands are
// read back from the ring transposed (lane = factor dimension, register = non-zero): the gather is coalesced and the
// transposition is free.  Index and value chunks travel through the same DMA path (indices 2(D-1) steps ahead).  All
// VMEM traffic of the kernel is issued from inline asm with counted s_waitcnt vmcnt(N): hipcc would otherwise drain the
// queue (vmcnt(0)) at every use and serialise gather and compute; row metadata comes through scalar loads (lgkmcnt).
#include <type_traits>
#include <utility>

#include "wrmf_internal.h"
#include "wrmf_device.h"
#include "wrmf_ldlt.h"

namespace rsparse_hip {
namespace {

using namespace dev;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef RSP_ABL
#define RSP_ABL 0   // dev builds only: ablations of the accumulate step (1 = no MFMA, 2 = no split, 4 = no look-ahead copies)
#endif
constexpr float kCgTolNe = 1e-10f;  // CG_TOL, inst/include/wrmf.hpp:22
constexpr int kStepNnz = 16;        // K of v_mfma_f32_32x32x16_bf16

// Which wave accumulates what.  Rank 128 with implicit feedback needs 20 accumulator tiles (320 registers): more than
// one wave can hold next to its operands, so there the waves work in PAIRS on the same steps (two ring groups instead
// of four) and split the tiles by role; everywhere else a wave owns every tile of its own steps.
//   NS = 3 (M1: 6 products per tile, M2: 3):  role 0 = M1 tile rows 2,3 + 2 M2 tiles (48 products, 9 tiles),
//                                             role 1 = M1 rows 0,1 + 8 M2 tiles (42 products, 11 tiles)
//   NS = 2 (3 and 3):                         role 0 = M1, role 1 = M2
struct NeMfmaOp { int slot, m, R, C, pa, pb; };   // acc[slot] += (m == 0 ? A-operand term pa : x term pa) of block R x (x term pb of block C)
struct NeUnit { int kind, t, pair; };             // a pair of floats to split: kind 0 = x, 1 = (c - 1) x; block t, floats 2 pair, 2 pair + 1

// SYM (implicit feedback, every confidence >= 1): both matrices are symmetric products of ONE operand set each,
//   M1 = a' a'^T with a' = sqrt(c - 1) s1 x,   M2 = (s2 x)(s2 x)^T     (s1, s2: powers of two, see ne_scales),
// split exactly into 2 fp16 terms (22 bits) with 3 products kept (2^-21 per product): half the matrix-core work of the
// bf16 path and, in a PAIR, half the split arithmetic per wave -- role 0 = M1 (needs only a'), role 1 = M2 + b (only x).
// QUAD (SYM at rank 128): all four waves of the workgroup consume the same steps, five tiles each.  80 accumulator
// registers per wave and one ring instead of two: the workgroup fits the CU twice, and the two resident workgroups
// overlap each other's matrix, vector, copy and per-row solve phases -- inside ONE wave they add up (measured: MFMA +
// split + copy issue = the step time).
// M2 = X_nnz X_nnz^T feeds ONLY the loss term y^T M2 y = sum_j (x_j.y)^2 (the system matrix is XtX + M1), so in the SYM
// path it is accumulated from the leading fp16 term alone (ONE product, 2^-11 per product, unbiased rounding: the row's
// loss moves by ~1e-5 relative, far inside the 1e-4 the tests hold the loss to; factors are untouched) while M1 keeps
// its three products (2^-21).  That is 40 instead of 60 MFMAs per step; the QUAD tiles are dealt so that every wave
// still owns five tiles and 9..11 MFMAs: wave 0 = M1 tiles 0-2 + M2 tiles 0-1, wave 1 = M1 3-5 + M2 2-3, wave 2 = M1 6-7 +
// M2 4-6 (and b, sum c: it reads every block of x anyway), wave 3 = M1 8-9 + M2 7-9.
template <int KP, int NS, bool IMPLICIT, bool SYM = false, bool QUAD = false>
struct NeRoles {
  static_assert(!SYM || (IMPLICIT && NS == 2), "SYM: implicit feedback, two fp16 terms");
  static_assert(!QUAD || (SYM && KP == 128), "QUAD: the fp16 path at rank 128");
  static constexpr int NB = KP / 32;
  static constexpr int NT = NB * (NB + 1) / 2;
  static constexpr bool PAIR = IMPLICIT && KP == 128 && !QUAD;
  static constexpr int NROLES = QUAD ? 4 : (PAIR ? 2 : 1);   // waves that consume the same steps
  static constexpr int NSETS = 4 / NROLES;                   // ring groups
  static constexpr int QH = 5;                               // QUAD: accumulator tiles per wave
  static constexpr int M2P = SYM ? 1 : 3;                    // products kept for the loss-only matrix M2 (see above)
  // QUAD: first M1 / M2 tile of wave w (tiles in the order of tile(R, C)); wave w owns [first(w), first(w + 1))
  __host__ __device__ static constexpr int q_first(int m, int w) {
    return m == 0 ? (w == 0 ? 0 : w == 1 ? 3 : w == 2 ? 6 : w == 3 ? 8 : 10) : (w == 0 ? 0 : w == 1 ? 2 : w == 2 ? 4 : w == 3 ? 7 : 10);
  }
  // Accumulator slot of tile (R, C) of matrix m (0 = the matrix of the system: implicit M1, explicit M2; 1 = implicit M2,
  // loss only) for a wave of the given role, or -1 if that wave does not accumulate the tile
  __host__ __device__ static constexpr int tile(int R, int C) { return R * (R + 1) / 2 + C; }
  __host__ __device__ static constexpr int slot(int role, int m, int R, int C) {
    if (!IMPLICIT) return m == 0 ? tile(R, C) : -1;
    if (QUAD) {
      const int t = tile(R, C);
      if (t < q_first(m, role) || t >= q_first(m, role + 1)) return -1;
      return m == 0 ? t - q_first(0, role) : (q_first(0, role + 1) - q_first(0, role)) + t - q_first(1, role);
    }
    if (!PAIR) return m * NT + tile(R, C);
    if (NS >= 3 && !SYM) {
      // role 0: M1 tile rows 2, 3 (slots 0..6, 42 products) + M2 tiles (0,0), (1,1) (slots 7, 8; 6 products)
      // role 1: M1 tile rows 0, 1 (slots 0..2, 18 products) + the other 8 M2 tiles (slots 3..10; 24 products)
      const bool m2_r0 = R == C && R < 2;
      if (role == 0) return m == 0 ? (R >= 2 ? tile(R, C) - 3 : -1) : (m2_r0 ? 7 + R : -1);
      if (m == 0) return R < 2 ? tile(R, C) : -1;
      if (m2_r0) return -1;
      return tile(R, C) == 1 ? 3 : tile(R, C) + 1;   // tiles 1, 3, 4, .., 9 -> 3, 4, 5, .., 10
    }
    return m == role ? tile(R, C) : -1;
  }
  static constexpr int NSLOT = !IMPLICIT ? NT : (QUAD ? QH : (!PAIR ? 2 * NT : (NS >= 3 && !SYM ? 11 : 10)));
  __host__ __device__ static constexpr bool owns_row(int role, int R) { return IMPLICIT && slot(role, 0, R, 0) >= 0; }
  __host__ __device__ static constexpr int rhs_role() { return QUAD ? 2 : (PAIR ? 1 : 0); }  // who accumulates b and sum c
  // QUAD: does wave `role` need block t of the operand a' = sqrt(c - 1) s1 x (m = 0) / of s2 x (m = 1)?
  __host__ __device__ static constexpr bool q_needs(int role, int m, int t) {
    for (int R = 0; R < NB; R++)
      for (int C = 0; C <= R; C++)
        if (slot(role, m, R, C) >= 0 && (R == t || C == t)) return true;
    return false;
  }

  // The MFMAs of one step in issue order: product-major, so that two MFMAs on the same accumulator are a whole sweep over
  // the tiles apart.  M2 (implicit) takes the products of its first two terms, the system matrix those of total order < NS.
  __host__ __device__ static constexpr int mfma_count(int role) {
    int n = 0;
    for (int R = 0; R < NB; R++)
      for (int C = 0; C <= R; C++) {
        if (slot(role, 0, R, C) >= 0) n += NS * (NS + 1) / 2;
        if (IMPLICIT && slot(role, 1, R, C) >= 0) n += M2P;
      }
    return n;
  }
  __host__ __device__ static constexpr NeMfmaOp mfma_op(int role, int j) {
    int n = 0;
    // the system matrix first: its (c - 1) x operands are then dead early and their registers serve the next step's
    for (int pa = 0; pa < NS; pa++)
      for (int pb = 0; pb + pa < NS; pb++)
        for (int R = 0; R < NB; R++)
          for (int C = 0; C <= R; C++)
            if (slot(role, 0, R, C) >= 0) {
              if (n == j) return NeMfmaOp{slot(role, 0, R, C), 0, R, C, pa, pb};
              n++;
            }
    if (IMPLICIT)
      for (int pa = 0; pa < (M2P == 1 ? 1 : 2); pa++)
        for (int pb = 0; pb + pa < (M2P == 1 ? 1 : 2); pb++)
          for (int R = 0; R < NB; R++)
            for (int C = 0; C <= R; C++)
              if (slot(role, 1, R, C) >= 0) {
                if (n == j) return NeMfmaOp{slot(role, 1, R, C), 1, R, C, pa, pb};
                n++;
              }
    return NeMfmaOp{-1, 0, 0, 0, 0, 0};
  }
  // The split work of one step: the 4 float pairs of every block of x, then those of (c - 1) x for the owned tile rows
  // (SYM pair: role 0 only a', role 1 only x)
  __host__ __device__ static constexpr int unit_count(int role) {
    if (QUAD) return 4 * (role % 2 == 0 ? 3 : 4);
    if (SYM && PAIR) return 4 * NB;
    int n = 4 * NB;
    for (int t = 0; t < NB; t++) n += owns_row(role, t) ? 4 : 0;
    return n;
  }
  __host__ __device__ static constexpr NeUnit unit(int role, int u) {
    if (QUAD) return NeUnit{role < 2 ? 1 : 0, u / 4, u % 4};
    if (SYM && PAIR) return NeUnit{role == 0 ? 1 : 0, u / 4, u % 4};
    if (u < 4 * NB) return NeUnit{0, u / 4, u % 4};
    int n = 4 * NB;
    for (int t = 0; t < NB; t++)
      if (owns_row(role, t)) {
        if (u < n + 4) return NeUnit{1, t, u - n};
        n += 4;
      }
    return NeUnit{-1, 0, 0};
  }
  // the unit before which the ring values of unit u's block must be requested: -1 = before the MFMA sequence
  __host__ __device__ static constexpr bool first_of_block(int role, int u) { return unit(role, u).pair == 0; }
};

template <int KP, int NROLES, bool IMPLICIT>
struct NeGeo {
  static constexpr bool PAIR = NROLES == 2, QUAD = NROLES == 4;
  static constexpr int NB = KP / 32;               // 32-wide blocks of the factor dimension
  static constexpr int NT = NB * (NB + 1) / 2;     // lower-triangular tiles
  static constexpr int VPI = 256 / KP;             // vectors per DMA instruction (64 lanes x 16 B)
  static constexpr int NI = kStepNnz / VPI;        // DMA instructions per step
  static constexpr int LPV = 64 / VPI;             // lanes per vector
  static constexpr int VEC_BYTES = NI * 1024;      // one step of gathered vectors
  static constexpr int SLOT_BYTES = VEC_BYTES + 256 /* values */;
  static constexpr int NRING = 4 / NROLES;
  static constexpr int D = KP == 128 ? 3 : 6;      // slots per ring; D - 1 steps requested ahead (a wave keeps only ~10 KB
                                                   // of LDS-DMA in flight -- measured, tools/probes -- so more does not help)
  static constexpr int IDXR = 2 * D - 1;           // index-chunk slots per ring (256 B each)
  static constexpr int LOADERS = NROLES;           // waves of a ring group that issue its DMA: all of them
  // DMA instructions per step and wave: its share of the vector pieces + the index chunk (first loader) and / or the
  // values (last loader)
  __host__ __device__ static constexpr int group_of(int role) {
    return NI / LOADERS + (role == 0 ? 1 : 0) + (role == LOADERS - 1 ? 1 : 0);
  }
  static constexpr int GROUP = NI / LOADERS + (LOADERS == 1 ? 2 : 1);   // the largest of them
  static constexpr int RING_BYTES = D * SLOT_BYTES + IDXR * 256;
  static constexpr int TLD = 33;                   // row stride inside a 32 x 32 tile (conflict-free by rows and by columns)
  static constexpr int A_FLOATS = NT * 32 * TLD;
  static constexpr int G_FLOATS = IMPLICIT && !QUAD ? A_FLOATS : 0;   // XtX as tiles, resident for the whole launch
                                                                     // (QUAD reads it from L2: half the LDS per workgroup)
  static constexpr int YB = QUAD ? 4 : 8;          // solved rows buffered in LDS before one wave stores them
  static constexpr int V_FLOATS = 4 * KP /* b partials */ + 2 * NB * NB * 32 /* matvec partials, double buffered */ +
                                  4 * KP /* per-wave published vector */ + 256 /* warm start (DMA target, 1 KB) */ +
                                  YB * KP + 4 * YB /* row ids, loss slots, losses (double) */ + 64 /* scalars */;
  static constexpr int BYTES = NRING * RING_BYTES + (A_FLOATS + G_FLOATS + V_FLOATS) * 4;
  static_assert(BYTES <= (QUAD ? 80 : 160) * 1024, "LDS budget");
  static_assert((D - 2) * GROUP <= 63, "vmcnt is a 6-bit field");
  __host__ __device__ static constexpr int tile(int R, int C) { return R * (R + 1) / 2 + C; }
};

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(reinterpret_cast<uintptr_t>(p));  // low 32 bits of a generic LDS pointer = LDS byte address
}

// ---- LDS-DMA, issued from asm so that the loop's s_waitcnt can be counted (see the header) ----
// One 16-byte piece per lane: LDS destination = M0 + lane * 16 (wave-uniform base), source = each lane's own pointer.
__device__ __forceinline__ void dma16(const void* g, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_base) : "memory", "m0");
}
// One dword per lane: LDS destination = M0 + lane * 4.
__device__ __forceinline__ void dma4(const void* g, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(g), "s"(lds_base) : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// scalar load (lgkmcnt, not vmcnt: does not disturb the DMA queue accounting); the address must be wave-uniform
__device__ __forceinline__ int sload(const int* p) {
  int v;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  return v;
}

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  const f32x2 f = {lo, hi};
  const bf16x2 h = __builtin_convertvector(f, bf16x2);  // v_cvt_pk_bf16_f32 (round to nearest even)
  return __builtin_bit_cast(unsigned, h);
}
// one term of the exact bf16 expansion of the pair r: returns the packed term and leaves the residuals (v_pk_add_f32)
__device__ __forceinline__ unsigned split_stage(f32x2& r, bool last) {
  const bf16x2 hb = __builtin_convertvector(r, bf16x2);  // v_cvt_pk_bf16_f32 (round to nearest even)
  const unsigned pk = __builtin_bit_cast(unsigned, hb);
  if (!last) {
    const f32x2 hi = {__uint_as_float(pk << 16), __uint_as_float(pk & 0xffff0000u)};
    r -= hi;
  }
  return pk;
}

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// the same with fp16 terms (v_cvt_pk_f16_f32, round to nearest even; x - fl16(x) is exact in fp32 whatever fl16 did)
__device__ __forceinline__ unsigned split_stage_h(f32x2& r, bool last) {
  const f16x2 hb = __builtin_convertvector(r, f16x2);
  const unsigned pk = __builtin_bit_cast(unsigned, hb);
  if (!last) {   // r - fl16(r) in one mixed-precision FMA per element (no widening conversions)
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(pk), "v"(r.x));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(pk), "v"(r.y));
    r = f32x2{r0, r1};
  }
  return pk;
}
// one packed multiply (hipcc scalarises the pair when its consumer is the mixed-precision FMA above)
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x16 mfma_f16(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// Powers of two that bring the operands of the SYM path into fp16 range: the largest |s2 x| and |s1 sqrt(c-1) x| land
// in [2^13, 2^15) (fp16 holds 65504; the terms of small entries may be subnormal, which costs absolute, not relative,
// accuracy: <= 2^-39 of the largest entry).  stats = {bits of max |x|, bits of max c, any c < 1} (ne_stats_kernel).
struct NeScales { float s1sq, s2, inv1, inv2; };
__device__ __forceinline__ NeScales ne_scales(const unsigned* stats) {
  const float mx = __uint_as_float(stats[0]), mc = __uint_as_float(stats[1]);
  const float ma = mx * __builtin_sqrtf(fmaxf(mc - 1.f, 0.f));
  auto pow2_for = [](float m) {
    int e = 0;
    if (m > 0.f && m < 3.0e38f) (void)frexpf(m, &e);   // m = f 2^e, f in [0.5, 1)
    e = max(-40, min(40, 14 - e));                     // m 2^(14 - e) < 2^14 ... rounding of sqrt and products: one bit spare
    return e;
  };
  const int e1 = pow2_for(ma), e2 = pow2_for(mx);
  auto uni = [](float v) { return __int_as_float(rfl(__float_as_int(v))); };   // wave-uniform: keep them in SGPRs
  return NeScales{uni(ldexpf(1.f, 2 * e1)), uni(ldexpf(1.f, e2)), uni(ldexpf(1.f, -2 * e1)), uni(ldexpf(1.f, -2 * e2))};
}

__device__ __forceinline__ f32x16 mfma_bf16(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0,
                                                 0);
}

__device__ __forceinline__ float row16_sum_ne(float v) {
  v += dpp<0xB1>(v);
  v += dpp<0x4E>(v);
  v += dpp<0x141>(v);
  v += dpp<0x140>(v);
  return v;
}
// sum over lanes l, l^16, l^32, l^48 (bitwise identical in all four)
__device__ __forceinline__ float groups_sum_ne(float v) {
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
__device__ __forceinline__ float half_swap_sum(float v) {  // v(l) + v(l ^ 32)
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float wave_sum_all(float v) { return groups_sum_ne(row16_sum_ne(v)); }
// LDS float add without return value (ds_add_f32)
__device__ __forceinline__ void lds_add(float* p, float v) {
  (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// position of a ring group's gather stream: row `li` of the workgroup's list, step `s` of the group's steps in that row
struct NeCursor {
  int li, s, nst, p1, cnt;
  bool live;
};

template <int NB, int NS>
struct NeParts {   // MFMA operands of one step: block t, term q
  u32x4 x[NB][NS];   // x
  u32x4 a[NB][NS];   // (c - 1) x   (tile rows this wave owns; implicit only)
};

// COLLECT: the second launch for the rows that were split across workgroups (see the publish step below): one workgroup
// per split row, no streaming -- it sums the published partial sums of the row's segments in segment order (the same
// per-wave register images, so every geometry above works unchanged) and runs the same per-row solve.
// GB (conjugate gradient, implicit feedback with a global bias: cg_solver_implicit_global_bias, wrmf_implicit.hpp:35-57):
// the first residual is  X_nnz (c - c1 % (X_nnz^T x + g)) - XtX x + base  =  b - A x + [base - g X_nnz (c - 1)];  the term
// in brackets comes per row from a.ne_r0 (launch_gb_row_terms, one more pass over these rows: a rare variant), the CG steps
// are unchanged, and the loss  sum_j c_j (tau - x_j.y)^2,  tau = 1 - g,  is  tau^2 sum c - 2 tau y.b + y^T (M1 + M2) y.
template <int KP, int NS, bool IMPLICIT, bool SYM, bool QUAD, bool COLLECT, bool CHOL, bool GB = false>
__global__ __launch_bounds__(256, QUAD ? 2 : 1) void als_ne_kernel(AlsArgs a, const int32_t* __restrict__ wg_rows,
                                                                    const int32_t* __restrict__ wg_ptr, int slot0,
                                                                    double* __restrict__ row_loss, int only_if_lt1) {
  using RL = NeRoles<KP, NS, IMPLICIT, SYM, QUAD>;
  using G_ = NeGeo<KP, RL::NROLES, IMPLICIT>;
  constexpr int NROLES = RL::NROLES;
  constexpr int NB = G_::NB, NT = G_::NT, NI = G_::NI, LPV = G_::LPV, D = G_::D, IDXR = G_::IDXR, TLD = G_::TLD;
  constexpr int NM = IMPLICIT ? 2 : 1;  // accumulated matrices: implicit {M1, M2}, explicit {M2}
  constexpr int NSETS = RL::NSETS, NSLOT = RL::NSLOT, YB = G_::YB;
  constexpr bool PAIR = RL::PAIR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = rfl(tid >> 6);
  const int h = lane >> 5, d = lane & 31;
  const int k = a.k;
  // Which of the two launches of an implicit half-iteration does the work is decided on the device (no host round trip):
  // the SYM kernel needs every confidence >= 1 (it takes sqrt(c - 1)), the bf16 kernel behind it runs only otherwise.
  NeScales scl = {1.f, 1.f, 1.f, 1.f};
  if constexpr (IMPLICIT) {
    if (a.ne_stats) {
      const int lt1 = sload(reinterpret_cast<const int*>(a.ne_stats) + 2);
      if (SYM ? lt1 != 0 : (only_if_lt1 != 0 && lt1 == 0)) return;
      if constexpr (SYM) {
        unsigned st[2];
        st[0] = (unsigned)sload(reinterpret_cast<const int*>(a.ne_stats));
        st[1] = (unsigned)sload(reinterpret_cast<const int*>(a.ne_stats) + 1);
        scl = ne_scales(st);
      }
    }
  }
  const int wset = wv / NROLES;   // ring group: its steps of a row are wset, wset + NSETS, ...
  const int wrole = wv % NROLES;
  // f(integral_constant<int, role of this wave>)
  auto with_role = [&](auto&& f) {
    if constexpr (NROLES == 1) {
      f(std::integral_constant<int, 0>{});
    } else if constexpr (NROLES == 2) {
      if (wrole == 0) f(std::integral_constant<int, 0>{});
      else f(std::integral_constant<int, 1>{});
    } else {
      if (wrole == 0) f(std::integral_constant<int, 0>{});
      else if (wrole == 1) f(std::integral_constant<int, 1>{});
      else if (wrole == 2) f(std::integral_constant<int, 2>{});
      else f(std::integral_constant<int, 3>{});
    }
  };

  char* ring = smem + wset * G_::RING_BYTES;
  const unsigned ring_a = rfl((int)lds_addr(ring));
  float* sA = reinterpret_cast<float*>(smem + G_::NRING * G_::RING_BYTES);  // NT tiles of 32 x TLD: the row's system
  float* sG = sA + G_::A_FLOATS;     // XtX in the same tile layout (implicit)
  float* sB = sG + G_::G_FLOATS;     // [4][KP]        right-hand-side partials of the waves
  float* sPart = sB + 4 * KP;        // [2][NB*NB][32] matrix-vector partials, double buffered
  float* sPub = sPart + 2 * NB * NB * 32;  // [4][KP]  per-wave published vector
  float* sX0 = sPub + 4 * KP;        // [256]          warm start of the row (DMA target)
  float* sY = sX0 + 256;             // [YB][KP]       solved rows waiting to be stored
  int* sYrow = reinterpret_cast<int*>(sY + YB * KP);          // [YB] row ids
  int* sYli = sYrow + YB;                                     // [YB] loss slots
  double* sYloss = reinterpret_cast<double*>(sYli + YB);      // [YB]
  float* sScal = reinterpret_cast<float*>(sYloss + YB);       // [64]

  const int dq = lane / LPV;                     // which vector of a DMA instruction this lane copies
  const int dl4 = min((lane % LPV) * 4, k - 4);  // its 16-byte piece; pieces beyond the rank re-read the last one
                                                 // (never past the vector) and are zeroed when they are consumed
  const int list_begin = sload(wg_ptr + blockIdx.x), list_end = sload(wg_ptr + blockIdx.x + 1);

  if constexpr (G_::G_FLOATS > 0) {   // XtX -> LDS tiles, once (ordinary loads: the DMA queue is still empty)
    for (int e = tid; e < NT * 1024; e += 256) {
      const int t = e >> 10, i = (e >> 5) & 31, j = e & 31;
      const int R = t >= 6 ? 3 : (t >= 3 ? 2 : (t >= 1 ? 1 : 0)), C = t - R * (R + 1) / 2;
      const int gi = 32 * R + i, gj = 32 * C + j;
      sG[t * 32 * TLD + i * TLD + j] = (gi < k && gj < k) ? a.XtX[(size_t)gi * k + gj] : 0.f;
    }
  }

  // List entry -> the run of non-zeros to stream.  entry >= 0: a whole row; -(s + 1): segment s of a row that the host
  // split across workgroups (wrmf_capi.cpp): {row, first non-zero, non-zeros, index, segments of the row, scratch slot}
  struct NeEntry { int row, p1, cnt, cnt_row, seg, nseg, slot; };
  auto decode = [&](const int li) {
    NeEntry en;
    const int e = sload(wg_rows + li);
    if (e >= 0) {
      en.row = e;
      en.p1 = sload(a.col_ptrs + e);
      en.cnt = en.cnt_row = sload(a.col_ptrs + e + 1) - en.p1;
      en.seg = 0; en.nseg = 1; en.slot = 0;
    } else {
      const int32_t* sg = a.ne_segs + (size_t)(-e - 1) * 6;
      en.row = sload(sg);
      const int rp = sload(a.col_ptrs + en.row);
      en.cnt_row = sload(a.col_ptrs + en.row + 1) - rp;
      en.p1 = rp + sload(sg + 1);
      en.cnt = sload(sg + 2);
      en.seg = sload(sg + 3); en.nseg = sload(sg + 4); en.slot = sload(sg + 5);
    }
    return en;
  };

  // ---- gather stream of this ring group ----
  auto load_row = [&](NeCursor& c) {
    c.s = 0;
    if (c.li < list_end) {
      const NeEntry en = decode(c.li);
      c.p1 = en.p1;
      c.cnt = en.cnt;
      const int nsteps = (c.cnt + kStepNnz - 1) / kStepNnz;
      c.nst = (nsteps - wset + NSETS - 1) / NSETS;
    } else {
      c.live = false;
      c.nst = 1 << 30;
    }
  };
  auto advance = [&](NeCursor& c) {
    c.s++;
    while (c.live && c.s >= c.nst) {
      c.li++;
      load_row(c);
    }
  };
  auto start = [&](NeCursor& c) {
    c.li = list_begin - 1;
    c.s = 0;
    c.nst = 0;
    c.live = true;
    c.p1 = 0;
    c.cnt = 1;
    advance(c);
  };
  NeCursor ci, cv;   // index-chunk stream, vector stream
  int islot = 0, vslot = 0, vislot = 0;  // ring slots: next index chunk, next vector step, index chunk that step reads
  auto issue_idx = [&]() {  // index chunk of the step (and the 48 entries behind it), clamped to the row
    const int pos = ci.live ? ci.p1 + min((wset + NSETS * ci.s) * kStepNnz + lane, ci.cnt - 1) : 0;
    if (wrole == 0) dma4(a.row_idx + pos, ring_a + (unsigned)(D * G_::SLOT_BYTES + islot * 256));
    islot = islot + 1 == IDXR ? 0 : islot + 1;
    advance(ci);
  };
  // needs the step's index chunk in LDS; past the end of the list: harmless copies of row 0.  In a pair each wave issues
  // half of the vector pieces (role 0 also the index chunks, role 1 the values)
  auto issue_vec = [&](auto role_tag) {
    constexpr int LR = decltype(role_tag)::value;
    const int* ix = reinterpret_cast<const int*>(ring + D * G_::SLOT_BYTES + vislot * 256) + dq * NI;
    int id[NI];
    if constexpr (NI == 8) {
      const int4 i0 = *reinterpret_cast<const int4*>(ix), i1 = *reinterpret_cast<const int4*>(ix + 4);
      id[0] = i0.x; id[1] = i0.y; id[2] = i0.z; id[3] = i0.w;
      id[4] = i1.x; id[5] = i1.y; id[6] = i1.z; id[7] = i1.w;
    } else {
      const int4 i0 = *reinterpret_cast<const int4*>(ix);
      id[0] = i0.x; id[1] = i0.y; id[2] = i0.z; id[3] = i0.w;
    }
    const unsigned base = ring_a + (unsigned)(vslot * G_::SLOT_BYTES);
#pragma unroll
    for (int e = 0; e < NI; e++)
      if (e / (NI / G_::LOADERS) == LR) dma16(a.X + (size_t)id[e] * k + dl4, base + e * 1024);
    const int pos = cv.live ? cv.p1 + min((wset + NSETS * cv.s) * kStepNnz + lane, cv.cnt - 1) : 0;
    if (LR == G_::LOADERS - 1) dma4(a.vals + pos, base + G_::VEC_BYTES);
    vslot = vslot + 1 == D ? 0 : vslot + 1;
    vislot = vislot + 1 == IDXR ? 0 : vislot + 1;
    advance(cv);
  };
  // The same two issues split into a preparation (addresses into registers, stream positions advanced) and the DMA
  // instructions themselves, which the accumulate loop spreads between its MFMAs: issued in one burst at the top of a
  // step they fill the wave's DMA queue (measured: ~4 B/clk per wave) and the wave stalls at the issue instead of
  // computing; spread out, every one finds room and the copy runs behind the arithmetic.
  const void* pre_src[G_::GROUP];   // per piece: this lane's source address
  unsigned pre_dst[G_::GROUP];      // ... wave-uniform LDS destination
  int pre_id[NI];                   // indices of the step being prepared (read from the ring ahead of their use)
  auto prepare_read = [&]() {
    const int* ix = reinterpret_cast<const int*>(ring + D * G_::SLOT_BYTES + vislot * 256) + dq * NI;
    if constexpr (NI == 8) {
      const int4 i0 = *reinterpret_cast<const int4*>(ix), i1 = *reinterpret_cast<const int4*>(ix + 4);
      pre_id[0] = i0.x; pre_id[1] = i0.y; pre_id[2] = i0.z; pre_id[3] = i0.w;
      pre_id[4] = i1.x; pre_id[5] = i1.y; pre_id[6] = i1.z; pre_id[7] = i1.w;
    } else {
      const int4 i0 = *reinterpret_cast<const int4*>(ix);
      pre_id[0] = i0.x; pre_id[1] = i0.y; pre_id[2] = i0.z; pre_id[3] = i0.w;
    }
  };
  auto prepare = [&](auto role_tag) {
    constexpr int LR = decltype(role_tag)::value;
    int n = 0;
    if (LR == 0) {
      const int pos = ci.live ? ci.p1 + min((wset + NSETS * ci.s) * kStepNnz + lane, ci.cnt - 1) : 0;
      pre_src[n] = a.row_idx + pos;
      pre_dst[n++] = ring_a + (unsigned)(D * G_::SLOT_BYTES + islot * 256);
    }
    islot = islot + 1 == IDXR ? 0 : islot + 1;
    advance(ci);
    const unsigned base = ring_a + (unsigned)(vslot * G_::SLOT_BYTES);
#pragma unroll
    for (int e = 0; e < NI; e++)
      if (e / (NI / G_::LOADERS) == LR) {
        pre_src[n] = a.X + (size_t)pre_id[e] * k + dl4;
        pre_dst[n++] = base + e * 1024;
      }
    if (LR == G_::LOADERS - 1) {
      const int pos = cv.live ? cv.p1 + min((wset + NSETS * cv.s) * kStepNnz + lane, cv.cnt - 1) : 0;
      pre_src[n] = a.vals + pos;
      pre_dst[n++] = base + G_::VEC_BYTES;
    }
    vslot = vslot + 1 == D ? 0 : vslot + 1;
    vislot = vislot + 1 == IDXR ? 0 : vislot + 1;
    advance(cv);
  };
  auto piece = [&](auto role_tag, auto nc) {   // piece n of the prepared step: 4-byte copies are the index / value chunks
    constexpr int LR = decltype(role_tag)::value, N = decltype(nc)::value;
    if constexpr (N < G_::group_of(LR)) {
      constexpr bool small = (LR == 0 && N == 0) || (LR == G_::LOADERS - 1 && N == G_::group_of(LR) - 1);
      if constexpr (small) dma4(pre_src[N], pre_dst[N]);
      else dma16(pre_src[N], pre_dst[N]);
    }
  };
  if constexpr (!COLLECT) {
    start(ci);
    start(cv);
    // steady state, consuming stream position q: issue {index chunk q + 2(D-1), vectors q + D-1}.  Prologue = the
    // index chunks 0 .. D-2, then the "virtual" positions -(D-1) .. -1
    for (int j = 0; j < D - 1; j++) issue_idx();
    wait_vm<0>();
    __syncthreads();   // the partner reads the index chunks too; XtX tiles are in place
    for (int j = 0; j < D - 1; j++) {
      issue_idx();
      with_role([&](auto rc) { issue_vec(rc); });
    }
  }
  int cslot = 0;  // ring slot of the next step this wave consumes
  int buf = 0;
  int nbuf = 0;   // solved rows waiting in sY
#ifdef RSP_NE_PROF
  unsigned long long prof_t[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long prof_start = __builtin_amdgcn_s_memtime();
  unsigned long long _tl = prof_start;
#define NE_T(j) { const unsigned long long _t1 = __builtin_amdgcn_s_memtime(); prof_t[j] += _t1 - _tl; _tl = _t1; }
#else
#define NE_T(j)
#endif
#ifndef RSP_NECH_ABL
#define RSP_NECH_ABL 0
#endif

  for (int li = list_begin; li < list_end; li++) {
    const NeEntry en = decode(li);
    const int row = en.row, p1 = en.p1, cnt = en.cnt;
    const int nsteps = (cnt + kStepNnz - 1) / kStepNnz;
    const int nst = (nsteps - wset + NSETS - 1) / NSETS;   // this group's steps
    const int nst_max = (nsteps + NSETS - 1) / NSETS;      // group 0's steps (barrier count in PAIR mode)
    const float lam_use = IMPLICIT ? 0.f : (float)(a.lambda_loss * (a.dynamic_lambda ? (double)(float)en.cnt_row : 1.0));
    // warm start -> LDS by DMA (an ordinary load would make hipcc drain the look-ahead queue where the value is used).
    // It is older than every vector group this row issues, so it has landed once a step issued in this row has been
    // waited for, i.e. after D - 1 steps of wave 0's group; shorter rows drain explicitly below
    if (wv == 0) dma16(a.Y + (size_t)row * k + min((lane & 31) * 4, k - 4), rfl((int)lds_addr(sX0)));
    NE_T(7)

    // accumulators: NSLOT tiles; what a slot holds depends on the wave's role (NeRoles::slot)
    f32x16 acc[NSLOT];
#pragma unroll
    for (int t = 0; t < NSLOT; t++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[t][e] = 0.f;
    f32x2 bp[NB], bp_hi[NB];   // (even, odd non-zero of a pair) summed separately, joined after the row
#pragma unroll
    for (int t = 0; t < NB; t++) bp[t] = bp_hi[t] = f32x2{0.f, 0.f};
    double sc = 0.0;

    // ---- accumulate.  Software-pipelined by one step: the MFMAs of step i-1 (operands `cur`, split in the previous
    // iteration) are issued one by one, each followed by a slice of the split arithmetic of step i (operands `nxt`),
    // which is independent of them.  The order is pinned with sched_barrier: left alone, hipcc issues the MFMAs back to
    // back and the ~300 VALU instructions after them, and the step takes the SUM of the two instead of the maximum.
    auto accumulate = [&](auto role_tag) {
      constexpr int ROLE = decltype(role_tag)::value;
      constexpr bool RHS = ROLE == RL::rhs_role();
      constexpr bool ANYRHS = RHS, SCR = RHS;
      constexpr int NMFMA = RL::mfma_count(ROLE), NU = RL::unit_count(ROLE), NHU = 2 * NU;
      constexpr int BARE = 4;   // MFMAs issued before the first split slice: they cover the latency of the ring reads
      using Parts = NeParts<NB, NS>;
      Parts pA, pB;
#pragma unroll
      for (int t = 0; t < NB; t++)
#pragma unroll
        for (int q = 0; q < NS; q++) pA.x[t][q] = pA.a[t][q] = pB.x[t][q] = pB.a[t][q] = u32x4{0u, 0u, 0u, 0u};
      auto mfma_j = [&](auto jc, Parts& cur) {
        constexpr auto op = RL::mfma_op(ROLE, decltype(jc)::value);
        if constexpr (SYM) {
          if constexpr (op.m == 0) acc[op.slot] = mfma_f16(cur.a[op.R][op.pa], cur.a[op.C][op.pb], acc[op.slot]);
          else acc[op.slot] = mfma_f16(cur.x[op.R][op.pa], cur.x[op.C][op.pb], acc[op.slot]);
        } else if constexpr (IMPLICIT && op.m == 0) {
          acc[op.slot] = mfma_bf16(cur.a[op.R][op.pa], cur.x[op.C][op.pb], acc[op.slot]);
        } else {
          acc[op.slot] = mfma_bf16(cur.x[op.R][op.pa], cur.x[op.C][op.pb], acc[op.slot]);
        }
      };
      // pipeline synchronisation of step i: returns false if this group has no step i (PAIR: barrier count)
      auto sync_step = [&](const int i) {
        const bool has = i < nst;   // PAIR mode: a group with one step fewer than group 0 still meets the barrier
        if (has) wait_vm<(D - 2) * G_::group_of(ROLE)>();  // this wave's share of the step has landed; D-2 later ones in flight
        NE_T(0)
        if constexpr (NROLES > 1) __builtin_amdgcn_s_barrier();  // ... and the partners' shares; the previous slot is released
        NE_T(1)
        if (has) prepare_read();   // the addresses are formed inside the step, behind its first MFMAs
        NE_T(2)
        return has;
      };
      // MFMAs on `cur` (step i-1) interleaved with the split of step i into `nxt`; masked = partial step or rank < KP
      auto step = [&](auto masked_tag, const int i, Parts& cur, Parts& nxt) {
        constexpr bool masked = decltype(masked_tag)::value;
        const int rem = cnt - (wset + NSETS * i) * kStepNnz;
        const char* slot = ring + cslot * G_::SLOT_BYTES;
        cslot = cslot + 1 == D ? 0 : cslot + 1;
        float c[8];      // values of this lane's 8 non-zeros (8h .. 8h+7)
        {
          const float4 c0 = *reinterpret_cast<const float4*>(slot + G_::VEC_BYTES + h * 32);
          const float4 c1 = *reinterpret_cast<const float4*>(slot + G_::VEC_BYTES + h * 32 + 16);
          c[0] = c0.x; c[1] = c0.y; c[2] = c0.z; c[3] = c0.w;
          c[4] = c1.x; c[5] = c1.y; c[6] = c1.z; c[7] = c1.w;
        }
        if constexpr (masked) {   // the padding slots of a row's last step hold copies of its last vector
#pragma unroll
          for (int e = 0; e < 8; e++) c[e] = (8 * h + e < rem) ? c[e] : 0.f;
        }
        // lane (h, d): dimension 32 t + d of its 8 non-zeros, one block of the factor dimension per unit group; the
        // values of a group are requested from the ring while the previous group is being split
        float raw[NU / 4][8];
        const char* lbase = slot + h * (8 / NI) * (KP * 4) + d * 4;
        auto load_group = [&](auto gc) {
          constexpr int GI = decltype(gc)::value;
          constexpr int t = RL::unit(ROLE, 4 * GI).t;
#pragma unroll
          for (int e = 0; e < 8; e++)
            raw[GI][e] = *reinterpret_cast<const float*>(lbase + (e % NI) * 1024 + (e / NI) * (KP * 4) + t * 128);
          if constexpr (masked) {   // ... and dimensions beyond the rank copies of the vector's last piece
#pragma unroll
            for (int e = 0; e < 8; e++) raw[GI][e] = (32 * t + d < k && 8 * h + e < rem) ? raw[GI][e] : 0.f;
          }
        };
        f32x2 ur[NU];   // residuals of the pairs between the two slices of a unit
        const f32x2 s2v = {scl.s2, scl.s2};
        f32x2 cm1[4];   // (c - 1) of the lane's 4 pairs; SYM: s1 sqrt(c - 1) (padding slots: c = 0 -> 0)
        if constexpr (SYM) {
#pragma unroll
          for (int q = 0; q < 4; q++)
            // s1 sqrt(c - 1) = sqrt(|s1^2 c - s1^2|): every real confidence is >= 1; the padding slots (c = 0, masked
            // steps only) multiply vectors that were zeroed
            cm1[q] = f32x2{__builtin_amdgcn_sqrtf(__builtin_fabsf(fmaf(c[2 * q], scl.s1sq, -scl.s1sq))),
                           __builtin_amdgcn_sqrtf(__builtin_fabsf(fmaf(c[2 * q + 1], scl.s1sq, -scl.s1sq)))};
        } else if constexpr (IMPLICIT) {
#pragma unroll
          for (int q = 0; q < 4; q++) cm1[q] = f32x2{c[2 * q] - 1.f, c[2 * q + 1] - 1.f};
        }
        auto half_unit = [&](auto hc) {
          constexpr int HU = decltype(hc)::value, U = HU / 2;
          constexpr auto un = RL::unit(ROLE, U);
          if constexpr (HU % 2 == 0) {
            if constexpr (U % 4 == 0 && U + 4 < NU) load_group(std::integral_constant<int, U / 4 + 1>{});
            f32x2 r = {raw[U / 4][2 * un.pair], raw[U / 4][2 * un.pair + 1]};
            if constexpr (un.kind == 0) {
              if constexpr (RHS) bp[un.t] += f32x2{c[2 * un.pair], c[2 * un.pair + 1]} * r;
              if constexpr (SYM) {
                r = pk_mul(r, s2v);
                nxt.x[un.t][0][un.pair] = split_stage_h(r, false);
              } else {
                nxt.x[un.t][0][un.pair] = split_stage(r, NS == 1);
              }
            } else {
              if constexpr (SYM) r = pk_mul(r, cm1[un.pair]);
              else r *= cm1[un.pair];
              nxt.a[un.t][0][un.pair] = SYM ? split_stage_h(r, false) : split_stage(r, NS == 1);
            }
            ur[U] = r;
          } else {
            f32x2 r = ur[U];
#pragma unroll
            for (int q = 1; q < NS; q++) {
              const unsigned pk = SYM ? split_stage_h(r, q == NS - 1) : split_stage(r, q == NS - 1);
              if constexpr (un.kind == 0) nxt.x[un.t][q][un.pair] = pk;
              else nxt.a[un.t][q][un.pair] = pk;
            }
          }
        };
        load_group(std::integral_constant<int, 0>{});
        // Sequence point after every (MFMA, split slice) pair: an empty asm that "redefines" the A operand of the next
        // MFMA and the input of the next slice.  Both are pure arithmetic, which neither sched_barrier nor program order
        // keeps in place (hipcc hoists all MFMAs above the splits); a fake dependence through a volatile asm does.
        auto pin = [&](auto jc) {
          constexpr int J = decltype(jc)::value;   // about to be issued: MFMA J and slice J - BARE
          constexpr int HU = J - BARE;
          constexpr auto op = RL::mfma_op(ROLE, J);
          u32x4& opnd = (IMPLICIT && op.m == 0) ? cur.a[op.R][op.pa] : cur.x[op.R][op.pa];
          if constexpr (HU >= 0 && HU < NHU) {
            constexpr auto un = RL::unit(ROLE, HU / 2);
            if constexpr (HU % 2 == 0) {
              float& inp = raw[HU / 8][2 * un.pair];
              asm volatile("" : "+v"(opnd), "+v"(inp));
            } else {
              f32x2& inp = ur[HU / 2];
              asm volatile("" : "+v"(opnd), "+v"(inp));
            }
          } else {
            asm volatile("" : "+v"(opnd));
          }
        };
        __builtin_amdgcn_sched_barrier(0);
        static_for<NMFMA>([&](auto jc) {
          constexpr int J = decltype(jc)::value;
          pin(jc);
#if !(RSP_ABL & 1)
          mfma_j(jc, cur);
#endif
          // (before the first look-ahead piece, which goes behind MFMA NMFMA / (GROUP + 1))
          if constexpr (J == (BARE - 1 < NMFMA / (G_::GROUP + 1) ? BARE - 1 : NMFMA / (G_::GROUP + 1)))
            prepare(std::integral_constant<int, ROLE>{});
#if !(RSP_ABL & 2)
          if constexpr (J >= BARE && J - BARE < NHU) half_unit(std::integral_constant<int, J - BARE>{});
#endif
          // piece n of the look-ahead copy behind MFMA (n + 1) NMFMA / (GROUP + 1)
          static_for<G_::GROUP>([&](auto nc) {
#if !(RSP_ABL & 4)
            if constexpr (J == (decltype(nc)::value + 1) * NMFMA / (G_::GROUP + 1))
              piece(std::integral_constant<int, ROLE>{}, nc);
#endif
          });
          __builtin_amdgcn_sched_barrier(0);
        });
#if !(RSP_ABL & 2)
        static_for<(NHU > NMFMA - BARE ? NHU - (NMFMA - BARE) : 0)>([&](auto hc) {
          half_unit(std::integral_constant<int, decltype(hc)::value + NMFMA - BARE>{});
        });
#endif
        if constexpr (RHS) {
          float s = 0.f;
#pragma unroll
          for (int e = 0; e < 8; e++) s += IMPLICIT ? c[e] : c[e] * c[e];
          sc += (double)s;
          // two-level sum of the right-hand side (bounds the fp32 running-sum error); arithmetic, not a branch
          const float fm = (i & 63) == 63 ? 1.f : 0.f;
#pragma unroll
          for (int t = 0; t < NB; t++) {
            bp_hi[t] += f32x2{fm, fm} * bp[t];
            bp[t] -= f32x2{fm, fm} * bp[t];
          }
        }
        NE_T(3)
      };
      // QUAD: no software pipeline and no pinned interleave -- the second workgroup on the CU fills this wave's gaps.
      // Split the step's operands, start the look-ahead copies, then the 15 MFMAs back to back.
      auto step_quad = [&](auto masked_tag, const int i) {
        constexpr bool masked = decltype(masked_tag)::value;
        const int rem = cnt - i * kStepNnz;
        const char* slot = ring + cslot * G_::SLOT_BYTES;
        cslot = cslot + 1 == D ? 0 : cslot + 1;
        float c[8];
        {
          const float4 c0 = *reinterpret_cast<const float4*>(slot + G_::VEC_BYTES + h * 32);
          const float4 c1 = *reinterpret_cast<const float4*>(slot + G_::VEC_BYTES + h * 32 + 16);
          c[0] = c0.x; c[1] = c0.y; c[2] = c0.z; c[3] = c0.w;
          c[4] = c1.x; c[5] = c1.y; c[6] = c1.z; c[7] = c1.w;
        }
        if constexpr (masked) {
#pragma unroll
          for (int e = 0; e < 8; e++) c[e] = (8 * h + e < rem) ? c[e] : 0.f;
        }
        const char* lbase = slot + h * (8 / NI) * (KP * 4) + d * 4;
        // the blocks of the factor dimension this wave touches: as a' = s1 sqrt(c - 1) x (its M1 tiles), as s2 x (its M2
        // tiles: leading fp16 term only) and, the right-hand-side wave, raw for b
        float raw[NB][8];
        static_for<NB>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          if constexpr (RL::q_needs(ROLE, 0, t) || RL::q_needs(ROLE, 1, t) || RHS) {
#pragma unroll
            for (int e = 0; e < 8; e++)
              raw[t][e] = *reinterpret_cast<const float*>(lbase + (e % NI) * 1024 + (e / NI) * (KP * 4) + t * 128);
            if constexpr (masked) {
#pragma unroll
              for (int e = 0; e < 8; e++) raw[t][e] = (32 * t + d < k && 8 * h + e < rem) ? raw[t][e] : 0.f;
            }
          }
        });
        prepare(std::integral_constant<int, ROLE>{});
        static_for<G_::GROUP>([&](auto nc) { piece(std::integral_constant<int, ROLE>{}, nc); });
        f32x2 mul[4];   // per pair: s1 sqrt(c - 1)
#pragma unroll
        for (int q = 0; q < 4; q++)
          mul[q] = f32x2{__builtin_amdgcn_sqrtf(__builtin_fabsf(fmaf(c[2 * q], scl.s1sq, -scl.s1sq))),
                         __builtin_amdgcn_sqrtf(__builtin_fabsf(fmaf(c[2 * q + 1], scl.s1sq, -scl.s1sq)))};
        const f32x2 s2v = {scl.s2, scl.s2};
        u32x4 opa[NB][2], opx[NB];
        static_for<NB>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const f32x2 r0 = {raw[t][2 * q], raw[t][2 * q + 1]};
            if constexpr (RHS) bp[t] += f32x2{c[2 * q], c[2 * q + 1]} * r0;
            if constexpr (RL::q_needs(ROLE, 0, t)) {
              f32x2 r = pk_mul(r0, mul[q]);
              opa[t][0][q] = split_stage_h(r, false);
              opa[t][1][q] = split_stage_h(r, true);
            }
            if constexpr (RL::q_needs(ROLE, 1, t)) {
              f32x2 r = pk_mul(r0, s2v);
              opx[t][q] = split_stage_h(r, true);
            }
          }
        });
        static_for<NMFMA>([&](auto jc) {
          constexpr auto o = RL::mfma_op(ROLE, decltype(jc)::value);
          if constexpr (o.m == 0) acc[o.slot] = mfma_f16(opa[o.R][o.pa], opa[o.C][o.pb], acc[o.slot]);
          else acc[o.slot] = mfma_f16(opx[o.R], opx[o.C], acc[o.slot]);
        });
        if constexpr (RHS) {
          float s = 0.f;
#pragma unroll
          for (int e = 0; e < 8; e++) s += c[e];
          sc += (double)s;
          const float fm = (i & 63) == 63 ? 1.f : 0.f;
#pragma unroll
          for (int t = 0; t < NB; t++) {
            bp_hi[t] += f32x2{fm, fm} * bp[t];
            bp[t] -= f32x2{fm, fm} * bp[t];
          }
        }
        NE_T(3)
      };
      const int niter = NROLES > 1 ? nst_max : nst;
      // only the row's last step can be partial, and it belongs to exactly one ring group; rank < KP: all masked
      const bool part = nst > 0 && cnt - (wset + NSETS * (nst - 1)) * kStepNnz < kStepNnz;
      const int nfull = k != KP ? 0 : (part ? nst - 1 : nst);
      if constexpr (QUAD) {
        for (int i = 0; i < nfull; i++) {
          sync_step(i);
          step_quad(std::false_type{}, i);
        }
        for (int i = nfull; i < niter; i++)
          if (sync_step(i)) step_quad(std::true_type{}, i);
      } else {
        for (int i = 0; i < nfull; i++) {
          sync_step(i);
          step(std::false_type{}, i, pA, pB);
          pA = pB;   // (two steps per trip with swapped operand sets would save these copies, but hipcc then moves the
        }            //  split arithmetic of the first step across the barrier and un-interleaves both)
        for (int i = nfull; i < niter; i++) {   // partial step, rank < KP, missing step of a shorter group
          if (sync_step(i)) {
            step(std::true_type{}, i, pA, pB);
            pA = pB;
          }
        }
        static_for<NMFMA>([&](auto jc) { mfma_j(jc, pA); });   // the last step
      }
      float bsum[NB];
#pragma unroll
      for (int t = 0; t < NB; t++) bsum[t] = 0.f;
      if constexpr (ANYRHS) {
#pragma unroll
        for (int t = 0; t < NB; t++) {
          const f32x2 v = bp_hi[t] + bp[t];
          bsum[t] = half_swap_sum(v.x + v.y);
        }
        sc += __shfl_xor(sc, 32);  // sc is uniform inside each half of the wave
      }
      if (lane < 32) {
#pragma unroll
        for (int t = 0; t < NB; t++) sB[wv * KP + 32 * t + lane] = bsum[t];
      }
      if (lane == 0) reinterpret_cast<double*>(sScal)[wv] = SCR ? sc : 0.0;
      NE_T(4)
    };
    // ---- A = XtX + M1 (explicit: lambda_use I + M2).  The tiles are dealt round robin to the ring groups: in phase ph
    // group g works on the tiles t with t % NSETS == (g + ph) % NSETS -- in phase 0 it writes G + its partial sums there,
    // in the later phases it adds its partial sums to what the groups before it left (read-add-write).  Every group is
    // busy in every phase, every tile sees the groups in a fixed order (deterministic); element (tile, e) of lane
    // (h, d) is row 8 (e / 4) + 4 h + e % 4, column d of the tile
    // QUAD (one ring group, every M1 tile has exactly one owner): XtX comes from L2 -- all of a wave's 80 loads in flight
    // together (plain loads: hipcc drains the queue for them, look-ahead copies included), then G + M1 straight into the tile
    auto chain_add_quad = [&](auto role_tag) {
      constexpr int ROLE = decltype(role_tag)::value;
      constexpr int T0 = RL::q_first(0, ROLE), TN = RL::q_first(0, ROLE + 1) - T0;   // this wave's M1 tiles (2 or 3)
      int h4 = 4 * h;
      asm volatile("" : "+v"(h4));   // keeps the offsets out of the row loop's preheader (they would be spilled)
      float g[TN][16];
      static_for<TN>([&](auto tc) {
        constexpr int T = T0 + decltype(tc)::value;
        constexpr int R = T >= 6 ? 3 : (T >= 3 ? 2 : (T >= 1 ? 1 : 0)), C = T - R * (R + 1) / 2;
        const int gj = min(32 * C + d, k - 1);
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int gi = min(32 * R + 8 * (e >> 2) + (e & 3) + h4, k - 1);
          g[decltype(tc)::value][e] = a.XtX[(unsigned)(gi * k + gj)];
        }
      });
      static_for<TN>([&](auto tc) {
        constexpr int TL = decltype(tc)::value, T = T0 + TL;
        constexpr int R = T >= 6 ? 3 : (T >= 3 ? 2 : (T >= 1 ? 1 : 0)), C = T - R * (R + 1) / 2;
        int toff = T * 32 * TLD + 4 * h * TLD + d;
        asm volatile("" : "+v"(toff));
        float* ta = sA + toff;
        const float cm = (32 * C + d < k) ? 1.f : 0.f;   // rank < 128: the padding of the tile is zero, not a clamped copy
        const f32x16 v = acc[TL] * scl.inv1;             // (slot of M1 tile T0 + TL = TL)
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const float rm = (32 * R + 8 * (e >> 2) + (e & 3) + h4 < k) ? cm : 0.f;
          ta[(8 * (e >> 2) + (e & 3)) * TLD] = fmaf(g[TL][e], rm, v[e]);
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    };
    auto chain_add = [&](auto role_tag, const int ph) {
      constexpr int ROLE = decltype(role_tag)::value;
      const bool first = ph == 0;
      const int mine = (wset + ph) % NSETS;
#pragma unroll
      for (int R = 0; R < NB; R++)
#pragma unroll
        for (int C = 0; C <= R; C++)
          if (RL::slot(ROLE, 0, R, C) >= 0 && G_::tile(R, C) % NSETS == mine) {
            // one address register per tile, opaque to hipcc: the element offsets then fit the 16-bit immediate of
            // the DS instructions (the tiles sit beyond 64 KB; without this it keeps one hoisted, spilled address
            // register per element)
            int toff = G_::tile(R, C) * 32 * TLD + 4 * h * TLD + d;
            asm volatile("" : "+v"(toff));
            float* ta = sA + toff;
            f32x16 v = acc[RL::slot(ROLE, 0, R, C)];
            if constexpr (SYM) v *= scl.inv1;
            if (first) {
              float g[16];
#pragma unroll
              for (int e = 0; e < 16; e++) {
                if constexpr (IMPLICIT) {
                  g[e] = (sG + toff)[(8 * (e >> 2) + (e & 3)) * TLD];
                } else {
                  g[e] = (R == C && 8 * (e >> 2) + (e & 3) + 4 * h == d) ? lam_use : 0.f;
                }
              }
#pragma unroll
              for (int e = 0; e < 16; e++) ta[(8 * (e >> 2) + (e & 3)) * TLD] = v[e] + g[e];
            } else {   // (plain read-add-write: ds_add_f32 measured ~450 cycles per wave instruction)
              float o[16];
#pragma unroll
              for (int e = 0; e < 16; e++) o[e] = ta[(8 * (e >> 2) + (e & 3)) * TLD];
#pragma unroll
              for (int e = 0; e < 16; e++) ta[(8 * (e >> 2) + (e & 3)) * TLD] = o[e] + v[e];
            }
          }
    };
    // ---- y^T (M1 + M2) y (explicit: y^T M2 y) from this wave's own accumulators; y is published in LDS (pub) and,
    // as x[], element 32 t + d in register t
    auto quad_form = [&](auto role_tag, const float* pub, const float (&x)[NB]) {
      constexpr int ROLE = decltype(role_tag)::value;
      float qf = 0.f;
#pragma unroll
      for (int R = 0; R < NB; R++) {
        float yi[16];
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) {
          const float4 v = *reinterpret_cast<const float4*>(pub + 32 * R + 8 * q4 + 4 * h);
          yi[4 * q4] = v.x; yi[4 * q4 + 1] = v.y; yi[4 * q4 + 2] = v.z; yi[4 * q4 + 3] = v.w;
        }
#pragma unroll
        for (int C = 0; C <= R; C++) {
          float s = 0.f;
#pragma unroll
          for (int m = 0; m < NM; m++)
            if (RL::slot(ROLE, m, R, C) >= 0) {
              float sm = 0.f;
#pragma unroll
              for (int e = 0; e < 16; e++) sm = fmaf(acc[RL::slot(ROLE, m, R, C)][e], yi[e], sm);
              s += SYM ? sm * (m == 0 ? scl.inv1 : scl.inv2) : sm;
            }
          qf = fmaf(R == C ? 1.f : 2.f, s * x[C], qf);
        }
      }
      return wave_sum_all(qf);
    };

    constexpr int PER_WAVE = kNeSegFloats / 4;
    static_assert(PER_WAVE >= NSLOT * 1024 + KP + 2, "scratch per wave");
    bool solved = true;
    if constexpr (!COLLECT) {
      with_role([&](auto rc) { accumulate(rc); });
      // ---- a segment of a row that the host split across workgroups: publish this wave's partial sums as they are
      // (accumulator registers, its b partial and sum c) and go on; the COLLECT launch adds them up and solves the row
      if (en.nseg > 1) {
        float* scr = a.ne_seg_scratch + (size_t)(en.slot + en.seg) * kNeSegFloats + wv * PER_WAVE;
        // (one opaque base pointer per tile and a compiler barrier between tiles: the element offsets fit the store's
        //  immediate field and nothing of this rare path is hoisted into the row loop's preheader)
#pragma unroll
        for (int t = 0; t < NSLOT; t++) {
          float* st = scr + t * 1024 + lane;
          asm volatile("" : "+v"(st) : : "memory");
#pragma unroll
          for (int e = 0; e < 16; e++) st[e * 64] = acc[t][e];
        }
        wave_sync();
        for (int e = lane; e < KP; e += 64) scr[NSLOT * 1024 + e] = sB[wv * KP + e];
        if (lane == 0) {
          *reinterpret_cast<double*>(scr + NSLOT * 1024 + KP) = reinterpret_cast<const double*>(sScal)[wv];
          if (wv == 0) row_loss[li - slot0] = 0.0;
        }
        solved = false;
      }
    } else {
      for (int s2 = 0; s2 < en.nseg; s2++) {
        const float* o = a.ne_seg_scratch + (size_t)(en.slot + s2) * kNeSegFloats + wv * PER_WAVE;
#pragma unroll
        for (int t = 0; t < NSLOT; t++) {
          const float* ot = o + t * 1024 + lane;
          asm volatile("" : "+v"(ot) : : "memory");
#pragma unroll
          for (int e = 0; e < 16; e++) acc[t][e] += ot[e * 64];
        }
        for (int e = lane; e < KP; e += 64) sB[wv * KP + e] = (s2 ? sB[wv * KP + e] : 0.f) + o[NSLOT * 1024 + e];
        wave_sync();
        if (lane == 0)
          reinterpret_cast<double*>(sScal)[wv] = (s2 ? reinterpret_cast<const double*>(sScal)[wv] : 0.0) +
                                                 *reinterpret_cast<const double*>(o + NSLOT * 1024 + KP);
      }
      wave_sync();
    }
    auto solve_row = [&]() __attribute__((always_inline)) {
    if (wv == 0 && (COLLECT || nst < D)) wait_vm<0>();   // short row: make sure the warm start has landed (see above)
    __syncthreads();   // warm start and every wave's right-hand-side partial are in LDS
    NE_T(12)
    for (int ph = 0; ph < NSETS; ph++) {
      if constexpr (QUAD) with_role([&](auto rc) { chain_add_quad(rc); });
      else with_role([&](auto rc) { chain_add(rc, ph); });
      NE_T(13)
      __syncthreads();
      NE_T(14)
    }
    NE_T(8)

    // ---- warm start, right-hand side and sum c, replicated in every wave; rank-vectors live as element 32 t + d in
    // register t (both halves of the wave alike)
    float x[NB], b[NB];
#pragma unroll
    for (int t = 0; t < NB; t++) {
      x[t] = (32 * t + d < k) ? sX0[32 * t + d] : 0.f;
      float s = sB[32 * t + d];
#pragma unroll
      for (int w2 = 1; w2 < 4; w2++) s += sB[w2 * KP + 32 * t + d];
      b[t] = s;
    }
    const double sc_row = (reinterpret_cast<const double*>(sScal)[0] + reinterpret_cast<const double*>(sScal)[1]) +
                          (reinterpret_cast<const double*>(sScal)[2] + reinterpret_cast<const double*>(sScal)[3]);

    // out = A v from the triangle.  NB^2 units of 32 x 32: unit (tg, sl) is the contribution of block sl of v to block
    // tg of the result -- sl >= tg: tile (sl, tg) read by rows (T^T v_sl), sl < tg: tile (tg, sl) read by columns; the
    // units go round robin to the 4 waves (rolled loop: the solve must stay small in the instruction cache), the two
    // halves of a wave split the 32-long inner product, partial blocks meet in LDS
    float* pub = sPub + wv * KP;
    auto matvec = [&](const float (&v)[NB], float (&out)[NB]) {
      wave_sync();
      if (h == 0) {
#pragma unroll
        for (int t = 0; t < NB; t++) pub[32 * t + d] = v[t];
      }
      wave_sync();
      float* part = sPart + buf * NB * NB * 32;
      NE_T(15)
      // the wave's NB^2 / 4 units together: all their LDS reads first (hipcc otherwise waits for each read before it
      // issues the next -- 16 round trips per unit), then the arithmetic
      constexpr int UW = NB * NB / 4, UB = QUAD ? 1 : UW;   // units in flight together (QUAD: 256 registers per wave, 80 of them accumulators)
#pragma unroll
      for (int i0 = 0; i0 < UW; i0 += UB) {
      float tv[UB][16], vv[UB][16];
#pragma unroll
      for (int i = 0; i < UB; i++) {
        const int u = wv + 4 * (i0 + i);
        const int tg = u / NB, sl = u % NB;
        const bool by_rows = sl >= tg;
        const int R = by_rows ? sl : tg, C = by_rows ? tg : sl;
        const float* T = sA + (R * (R + 1) / 2 + C) * 32 * TLD + (by_rows ? d + 16 * h * TLD : d * TLD + 16 * h);
        const int sv = by_rows ? TLD : 1;
        const float* vb = pub + 32 * sl + 16 * h;
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) {
          const float4 v4 = *reinterpret_cast<const float4*>(vb + 4 * q4);
          vv[i][4 * q4] = v4.x; vv[i][4 * q4 + 1] = v4.y; vv[i][4 * q4 + 2] = v4.z; vv[i][4 * q4 + 3] = v4.w;
        }
#pragma unroll
        for (int q = 0; q < 16; q++) tv[i][q] = T[q * sv];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < UB; i++) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) {
          s0 = fmaf(tv[i][4 * q4 + 0], vv[i][4 * q4 + 0], s0);
          s1 = fmaf(tv[i][4 * q4 + 1], vv[i][4 * q4 + 1], s1);
          s2 = fmaf(tv[i][4 * q4 + 2], vv[i][4 * q4 + 2], s2);
          s3 = fmaf(tv[i][4 * q4 + 3], vv[i][4 * q4 + 3], s3);
        }
        const float s = half_swap_sum((s0 + s1) + (s2 + s3));
        if (h == 0) part[(wv + 4 * (i0 + i)) * 32 + d] = s;
      }
      __builtin_amdgcn_sched_barrier(0);
      }
      NE_T(16)
      __syncthreads();
      NE_T(17)
#pragma unroll
      for (int t = 0; t < NB; t++) {
        float s = part[(t * NB) * 32 + d];
#pragma unroll
        for (int sl = 1; sl < NB; sl++) s += part[(t * NB + sl) * 32 + d];
        out[t] = s;
      }
      buf ^= 1;
      NE_T(18)
    };
    auto dot = [&](const float (&u)[NB], const float (&v)[NB]) {
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < NB; t++) s = fmaf(u[t], v[t], s);
      return wave_sum_all(s) * 0.5f;  // the two halves of the wave hold identical values
    };

    bool row_bad = false;   // CHOL: a non-positive pivot (known to every wave; wave 0 writes the loss)
    if constexpr (CHOL) {
      // solver == CHOLESKY (wrmf_implicit.hpp:231,236 / wrmf_explicit.hpp:103-108: y = solve(lhs, rhs)) on the system
      // assembled above: LDL^T with the matrix in the waves' registers (wrmf_ldlt.h; its scratch overlays the tiles, which
      // are dead once every wave has its columns).  A pivot that is not positive sends the row to the general solver
      // (wrmf_lu.hip), which also owns its loss term.
      using LD = Ldlt<KP, TLD>;
      static_assert(LD::FLOATS <= G_::A_FLOATS, "the LDL^T scratch overlays the tiles");
      float* sU = sPub;          // [KP] right-hand side -> solution
      if (tid >= k && tid < KP) sA[((tid >> 5) * ((tid >> 5) + 1) / 2 + (tid >> 5)) * 32 * TLD + (tid & 31) * (TLD + 1)] = 1.f;   // padding: unit diagonal
      if (wv == 0 && h == 0) {
#pragma unroll
        for (int t = 0; t < NB; t++) sU[32 * t + d] = b[t];
      }
      __syncthreads();
      NE_T(15)
      row_bad = LD::solve(sA, sA, sU, reinterpret_cast<int*>(sScal + 32), wv, lane);
      if (row_bad && tid == 0) {
        const int pos = atomicAdd(a.fail_counter, 1);
        if (pos < a.fail_cap) a.fail_rows[pos] = row;
      }
      NE_T(16)
#pragma unroll
      for (int t = 0; t < NB; t++) x[t] = (32 * t + d < k) ? sU[32 * t + d] : 0.f;
      __syncthreads();   // sPub is per-wave scratch again below
      NE_T(18)
    } else {
    // cg_solver_implicit / cg_solver_explicit on the assembled operator (one matrix-vector product per pass: pass 0
    // forms r = b - A x); rsold / alpha in double like the reference (wrmf_implicit.hpp:18)
    float r[NB], p[NB], ap[NB];
#pragma unroll
    for (int t = 0; t < NB; t++) p[t] = x[t];
    float r0x[NB];   // GB: base - g X_nnz (c - 1) of this row
#pragma unroll
    for (int t = 0; t < NB; t++) r0x[t] = 0.f;
    if constexpr (GB) {
      const int sl = sload(a.ne_r0_slot + row);
#pragma unroll
      for (int t = 0; t < NB; t++) r0x[t] = (32 * t + d < k) ? a.ne_r0[(size_t)sl * k + 32 * t + d] : 0.f;
    }
    double rsold = 0.0;
    bool conv = false;
    for (int it = 0; it <= a.cg_steps; ++it) {
      matvec(p, ap);  // uniform control flow: every wave runs the barrier inside
      if (it == 0) {
#pragma unroll
        for (int t = 0; t < NB; t++) p[t] = r[t] = (b[t] - ap[t]) + r0x[t];
        rsold = (double)dot(r, r);
      } else if (!conv) {
        const float alpha = (float)(rsold / (double)dot(p, ap));
#pragma unroll
        for (int t = 0; t < NB; t++) {
          x[t] = fmaf(alpha, p[t], x[t]);
          r[t] = fmaf(-alpha, ap[t], r[t]);
        }
        const double rsnew = (double)dot(r, r);
        if (rsnew < (double)kCgTolNe) {
          conv = true;
        } else {
          const float beta = (float)(rsnew / rsold);
#pragma unroll
          for (int t = 0; t < NB; t++) p[t] = fmaf(p[t], beta, r[t]);
          rsold = rsnew;
        }
      }
    }
    }
    NE_T(9)

    // ---- loss of the row: sum c - 2 y.b + y^T (M1 + M2) y + lambda |y|^2  (explicit: sum r^2 - 2 y.b + y^T M2 y)
    wave_sync();
    if (h == 0) {
#pragma unroll
      for (int t = 0; t < NB; t++) pub[32 * t + d] = x[t];
    }
    wave_sync();
    float qf = 0.f;
    with_role([&](auto rc) { qf = quad_form(rc, pub, x); });
    NE_T(10)
    if (lane == 0) sScal[16 + wv] = qf;
    const float yb = dot(x, b), yy = dot(x, x);
    __syncthreads();
    if (wv == 0) {   // park the solved row in LDS; stores happen YB rows at a time
      const double q = ((double)sScal[16] + (double)sScal[17]) + ((double)sScal[18] + (double)sScal[19]);
      const double tau = GB ? (double)a.loss_tgt_const : 1.0;
      const double fit = tau * tau * sc_row - 2.0 * tau * (double)yb + q;
      const double reg = IMPLICIT ? a.lambda_loss * (double)yy : (double)(lam_use * yy);
      if (lane == 0) {
        sYloss[nbuf] = row_bad ? 0.0 : fit + reg;
        sYrow[nbuf] = row;
        sYli[nbuf] = li;
      }
      if (h == 0) {
#pragma unroll
        for (int t = 0; t < NB; t++) sY[nbuf * KP + 32 * t + d] = x[t];
      }
    }
    };
    if (solved) {
      solve_row();
      nbuf++;
    }
    __syncthreads();  // A, b and the scalars are rewritten by the next row; the parked row is visible
    if (nbuf == YB || li + 1 == list_end) {
      // One wave stores the parked rows.  Stores count in vmcnt too and are not ordered with loads, so it drains its
      // queue afterwards: its counted waits must only ever see DMA (the look-ahead copies are older and have landed)
      if (wv == 3) {
        for (int rb = 0; rb < nbuf; rb++) {
          float* yr = a.Y + (size_t)sYrow[rb] * k;
          for (int e = lane; e < k; e += 64) yr[e] = sY[rb * KP + e];
          if (lane == 0) row_loss[sYli[rb] - slot0] = sYloss[rb];
        }
        wait_vm<0>();
      }
      nbuf = 0;
    }
    NE_T(11)
  }
  wait_vm<0>();  // the look-ahead copies issued past the end of the list
#ifdef RSP_NE_PROF
  if (a.ne_prof && lane == 0) {
    prof_t[5] = __builtin_amdgcn_s_memtime() - prof_start;
    prof_t[6] = list_end - list_begin;
    for (int j = 0; j < 20; j++) a.ne_prof[((size_t)blockIdx.x * 4 + wv) * 20 + j] = prof_t[j];
  }
#endif
}

template <int KP, int NS, bool IMPLICIT, bool SYM, bool QUAD, bool COLLECT, bool CHOL, bool GB = false>
hipError_t launch_ne_t(const AlsArgs& a, const int32_t* wg_rows, const int32_t* wg_ptr, int grid, double* row_loss,
                       hipStream_t s, int only_if_lt1, hipEvent_t* ev_slot = nullptr) {
  auto kern = als_ne_kernel<KP, NS, IMPLICIT, SYM, QUAD, COLLECT, CHOL, GB>;
  constexpr int lds = NeGeo<KP, NeRoles<KP, NS, IMPLICIT, SYM, QUAD>::NROLES, IMPLICIT>::BYTES;
  hipError_t err =
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (err != hipSuccess) return err;
  if (!COLLECT && !only_if_lt1) prof_note(ev_slot, reinterpret_cast<const void*>(kern));   // the launch that does the work
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a, wg_rows, wg_ptr, 0, row_loss, only_if_lt1);
  return hipGetLastError();
}

__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp<0xB1>(v));
  v = fmaxf(v, dpp<0x4E>(v));
  v = fmaxf(v, dpp<0x141>(v));
  v = fmaxf(v, dpp<0x140>(v));
  return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}

// max |x| over the fixed side, max c and "some c < 1" over the confidences -> stats[0..2] (zeroed by the caller).
// Non-negative floats order like their bit patterns, so one atomicMax per workgroup does it; a negative c only sets the flag.
__global__ __launch_bounds__(256) void ne_stats_kernel(const float* __restrict__ X, int64_t nx, const float* __restrict__ vals,
                                                        int64_t nnz, unsigned* __restrict__ stats,
                                                        const float* __restrict__ absmax_hint) {
  float mx = 0.f, mc = 0.f;
  if (absmax_hint) {   // the caller knows max |X| (rsparse_hip_hint_factor_absmax): nothing of X is read
    nx = 0;
    mx = fabsf(*absmax_hint);
  }
  int lt1 = 0;
  const int64_t stride = (int64_t)gridDim.x * 256 * 4, t0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  for (int64_t e = t0; e + 3 < nx; e += stride) {
    const float4 v = *reinterpret_cast<const float4*>(X + e);
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  for (int64_t e = (nx & ~(int64_t)3) + (int64_t)blockIdx.x * 256 + threadIdx.x; e < nx; e += (int64_t)gridDim.x * 256)
    mx = fmaxf(mx, fabsf(X[e]));
  {   // 16-byte loads over the aligned middle of vals, single floats at its two ends
    const int64_t head = min(nnz, (int64_t)((16 - (reinterpret_cast<uintptr_t>(vals) & 15)) & 15) / 4);
    const int64_t nvec = (nnz - head) / 4;
    const float4* v4 = reinterpret_cast<const float4*>(vals + head);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nvec; e += (int64_t)gridDim.x * 256) {
      const float4 c = v4[e];
      mc = fmaxf(fmaxf(mc, fmaxf(c.x, c.y)), fmaxf(c.z, c.w));
      lt1 |= !(c.x >= 1.f) | !(c.y >= 1.f) | !(c.z >= 1.f) | !(c.w >= 1.f);
    }
    if (blockIdx.x == 0 && threadIdx.x < 8) {   // at most 3 leading and 3 trailing values
      const int64_t tail0 = head + 4 * nvec;
      const int64_t e = threadIdx.x < 4 ? (int64_t)threadIdx.x : tail0 + (threadIdx.x - 4);
      const bool mine = threadIdx.x < 4 ? e < head : e < nnz;
      if (mine) {
        const float c = vals[e];
        mc = fmaxf(mc, c);
        lt1 |= !(c >= 1.f);
      }
    }
  }
  mx = wave_max(mx);
  mc = wave_max(mc);
  lt1 = __any(lt1);
  if ((threadIdx.x & 63) == 0) {
    if (!(mx < 3.0e38f)) mx = 3.0e38f;   // inf / nan in the factors: the solve will report it, the scales stay finite
    atomicMax(stats, __float_as_uint(mx));
    atomicMax(stats + 1, __float_as_uint(fmaxf(mc, 0.f)));
    if (lt1) atomicOr(stats + 2, 1u);
  }
}

// out[r][t] = base[t] - g * sum_j (c_j - 1) X[t, idx_j]  for row rows[r]; slot_of_row[rows[r]] = r.  One 256-thread workgroup
// per row.  A lane reads FOUR coordinates of a vector (16 bytes), KPW / 4 lanes a whole vector, so the workgroup's PARTS = 1024 / KPW
// lane groups take every PARTS-th non-zero each; eight non-zeros per trip and group, their indices first and their vectors in
// flight together; the groups' sums meet in LDS in a fixed order.  (Round 6.  Through round 5: a thread per coordinate, one
// non-zero per trip -- index, then vector, then the next index: two dependent round trips per non-zero, 22 ms per call at 1M x 100k
// where the same rows' conjugate-gradient launch takes 2.4; eight per trip with 4-byte reads: 9.3 ms.)
template <int KPW>
__global__ __launch_bounds__(256) void gb_row_terms_kernel(AlsArgs a, const int32_t* __restrict__ rows, int n,
                                                            float* __restrict__ out, int32_t* __restrict__ slot_of_row) {
  constexpr int LPR = KPW / 4, PARTS = 256 / LPR, U = 8;
  __shared__ float4 part[PARTS][LPR];
  const int r = blockIdx.x;
  if (r >= n) return;
  const int row = rows[r], k = a.k;
  const int c4 = threadIdx.x % LPR, pt = threadIdx.x / LPR;
  const int tc = min(4 * c4, k - 4);   // (k % 4 == 0: ne_supported)
  const int p1 = a.col_ptrs[row], p2 = a.col_ptrs[row + 1];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j0 = p1 + pt; j0 < p2; j0 += U * PARTS) {
    int id[U];
    float cv[U];
    float4 xv[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int j = j0 + u * PARTS;
      const bool ok = j < p2;
      id[u] = a.row_idx[ok ? j : p2 - 1];
      cv[u] = ok ? a.vals[j] - 1.f : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; u++) xv[u] = *reinterpret_cast<const float4*>(a.X + (size_t)id[u] * k + tc);
#pragma unroll
    for (int u = 0; u < U; u++) {
      acc.x = fmaf(cv[u], xv[u].x, acc.x);
      acc.y = fmaf(cv[u], xv[u].y, acc.y);
      acc.z = fmaf(cv[u], xv[u].z, acc.z);
      acc.w = fmaf(cv[u], xv[u].w, acc.w);
    }
  }
  part[pt][c4] = acc;
  __syncthreads();
  if (pt == 0 && 4 * c4 < k) {
    float4 sum = part[0][c4];
#pragma unroll
    for (int q = 1; q < PARTS; q++) {
      const float4 v = part[q][c4];
      sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    const float* base = a.rhs_init + 4 * c4;   // (k floats somewhere in a scratch buffer: no alignment promise)
    float* o = out + (size_t)r * k + 4 * c4;
    o[0] = base[0] - a.gbias * sum.x;
    o[1] = base[1] - a.gbias * sum.y;
    o[2] = base[2] - a.gbias * sum.z;
    o[3] = base[3] - a.gbias * sum.w;
  }
  if (threadIdx.x == 0) slot_of_row[row] = r;
}

}  // namespace

hipError_t launch_gb_row_terms(const AlsArgs& a, const int32_t* rows, int n, float* out, int32_t* slot_of_row, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (a.k <= 64) hipLaunchKernelGGL(gb_row_terms_kernel<64>, dim3(n), dim3(256), 0, s, a, rows, n, out, slot_of_row);
  else hipLaunchKernelGGL(gb_row_terms_kernel<128>, dim3(n), dim3(256), 0, s, a, rows, n, out, slot_of_row);
  return hipGetLastError();
}

bool ne_supported(int k) { return k > 32 && k <= 128 && k % 4 == 0; }

// wg_rows / wg_ptr: per-workgroup row lists (host-balanced, wrmf_capi.cpp build_ne_lists); row_loss: one double per
// entry of wg_rows
__global__ void ne_stats_words_kernel(unsigned* __restrict__ dst, const unsigned* __restrict__ src) {
  if (threadIdx.x < 2) dst[threadIdx.x] = src[threadIdx.x];
}
hipError_t launch_ne_stats(const float* X, int64_t nx, const float* vals, int64_t nnz, unsigned* stats, hipStream_t s,
                           const float* absmax_hint, const unsigned* cached_vstats, unsigned* save_vstats) {
  hipError_t err = hipMemsetAsync(stats, 0, 4 * sizeof(unsigned), s);
  if (err != hipSuccess) return err;
  // (frozen values whose statistics are known: only X is looked at -- or nothing, when the caller also knows max |X|: one small grid)
  const bool skip_vals = cached_vstats != nullptr;
  const int grid = (skip_vals && absmax_hint) ? 1 : 2048;
  hipLaunchKernelGGL(ne_stats_kernel, dim3(grid), dim3(256), 0, s, X, nx, vals, skip_vals ? (int64_t)0 : nnz, stats, absmax_hint);
  if ((err = hipGetLastError()) != hipSuccess) return err;
  if (skip_vals) hipLaunchKernelGGL(ne_stats_words_kernel, dim3(1), dim3(64), 0, s, stats + 1, cached_vstats);
  else if (save_vstats) hipLaunchKernelGGL(ne_stats_words_kernel, dim3(1), dim3(64), 0, s, save_vstats, stats + 1);
  return hipGetLastError();
}

// Implicit feedback with a.ne_stats: the fp16 SYM kernel and, behind it, the bf16 kernel that takes over when some
// confidence is below 1 (exactly one of the two does the work; the other returns at once).  Explicit feedback, or no
// stats: the bf16 kernel alone.
template <bool COLLECT, bool CHOL, bool GB = false>
hipError_t launch_ne_mode(const AlsArgs& a, const int32_t* wg_rows, const int32_t* wg_ptr, int n_wg, bool implicit,
                          double* row_loss, hipStream_t s, hipEvent_t* ev_slot = nullptr) {
  const int KP = padded_rank(a.k);
  hipError_t err;
#define RSP_NE_DISPATCH(KPV)                                                                                           \
  if (KP == KPV) {                                                                                                     \
    if constexpr (!GB) {                                                                                               \
      if (!implicit) return launch_ne_t<KPV, 3, false, false, false, COLLECT, CHOL>(a, wg_rows, wg_ptr, n_wg, row_loss, s, 0, ev_slot); \
      if (!a.ne_stats) return launch_ne_t<KPV, 3, true, false, false, COLLECT, CHOL>(a, wg_rows, wg_ptr, n_wg, row_loss, s, 0, ev_slot); \
    }                                                                                                                  \
    if (!implicit || !a.ne_stats) return hipErrorInvalidValue;   /* GB: implicit feedback with the value statistics */  \
    if ((err = launch_ne_t<KPV, 2, true, true, KPV == 128, COLLECT, CHOL, GB>(a, wg_rows, wg_ptr, n_wg, row_loss, s, 0, ev_slot)) !=  \
        hipSuccess)                                                                                                    \
      return err;                                                                                                      \
    return launch_ne_t<KPV, 3, true, false, false, COLLECT, CHOL, GB>(a, wg_rows, wg_ptr, n_wg, row_loss, s, 1);             \
  }
  RSP_NE_DISPATCH(128)
  RSP_NE_DISPATCH(64)
#undef RSP_NE_DISPATCH
  return hipErrorInvalidValue;
}

// The long rows of one half-iteration: the streaming launch over the per-workgroup lists (q.ne_rows / ne_ptr; an entry is
// a row or a segment of a split row), then, if rows were split, the COLLECT launch (one workgroup per split row).
// row_loss: one double per list entry, then one per split row.
hipError_t launch_als_ne(const AlsArgs& a, const QSchedule& q, bool implicit, double* row_loss, hipStream_t s,
                         hipEvent_t* ev_slot) {
  if (q.ne_wg <= 0) return hipSuccess;
  // a.ne_chol (solver == CHOLESKY): the instantiations whose per-row solve is the blocked LDL^T instead of CG -- separate
  // kernels, so neither solve costs the other's streaming loops a register
  const bool gb = !a.ne_chol && implicit && a.gbias != 0.f;   // conjugate gradient with a global bias
  if (gb && !(a.ne_r0 && a.ne_r0_slot)) return hipErrorInvalidValue;
  hipError_t err = a.ne_chol ? launch_ne_mode<false, true>(a, q.ne_rows, q.ne_ptr, q.ne_wg, implicit, row_loss, s, ev_slot)
                   : gb      ? launch_ne_mode<false, false, true>(a, q.ne_rows, q.ne_ptr, q.ne_wg, implicit, row_loss, s, ev_slot)
                             : launch_ne_mode<false, false>(a, q.ne_rows, q.ne_ptr, q.ne_wg, implicit, row_loss, s, ev_slot);
  if (err != hipSuccess || q.ne_nsplit <= 0) return err;
  return a.ne_chol ? launch_ne_mode<true, true>(a, q.ne_split_rows, q.ne_split_ptr, q.ne_nsplit, implicit, row_loss + q.ne_entries, s)
         : gb      ? launch_ne_mode<true, false, true>(a, q.ne_split_rows, q.ne_split_ptr, q.ne_nsplit, implicit, row_loss + q.ne_entries, s)
                   : launch_ne_mode<true, false>(a, q.ne_split_rows, q.ne_split_ptr, q.ne_nsplit, implicit, row_loss + q.ne_entries, s);
}

}  // namespace rsparse_hip
