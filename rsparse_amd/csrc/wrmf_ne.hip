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
// same operator, evaluated from the assembled matrix -- or the exact solve of the Cholesky branch.  The loss
// sum_j c_j (1 - x_j.y)^2  (wrmf_implicit.hpp:259-261) needs no second pass either: it equals
// sum c - 2 y.b + y^T (M1 + M2) y, and the quadratic forms are read off the accumulators.
//
// Arithmetic: fp32 MFMA runs at the vector rate (157 TF), bf16 MFMA 16x faster.  Every fp32 operand is split exactly
// into NS bf16 terms (x = x1 + x2 [+ x3], residuals are exact in fp32) and the products of total order < NS are
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16: NS = 3 keeps 6 products and is accurate to ~2^-23 per product, i.e.
// fp32-equivalent; M2 only feeds the loss and always uses the 3 products of its first two terms (2^-16).
//
// Layout: a workgroup = 4 waves = one row at a time; the waves take the row's 16-non-zero steps round robin, each with
// its own accumulators (lower-triangular 32x32 tiles: 10 per matrix at rank 128) and its own 3-deep LDS ring that
// LDS-DMA (global_load_lds_dwordx4, whole 512-byte vectors, no staging registers) fills two steps ahead; the MFMA
// operands are read back from the ring transposed (lane = factor dimension, register = non-zero), so the gather is
// coalesced and the transposition is free.  Index and value chunks travel through the same DMA path.  All VMEM traffic
// of the loop is issued from inline asm with counted s_waitcnt vmcnt(N): hipcc would otherwise drain the queue
// (vmcnt(0)) at every use and serialise gather and compute.
#include <type_traits>

#include "wrmf_internal.h"
#include "wrmf_device.h"

namespace rsparse_hip {
namespace {

using namespace dev;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr float kCgTolNe = 1e-10f;  // CG_TOL, inst/include/wrmf.hpp:22
constexpr int kRing = 3;            // vector / value slots per wave
constexpr int kStepNnz = 16;        // K of v_mfma_f32_32x32x16_bf16

template <int KP>
struct NeGeo {
  static constexpr int NB = KP / 32;               // 32-wide blocks of the factor dimension
  static constexpr int NT = NB * (NB + 1) / 2;     // lower-triangular tiles
  static constexpr int VPI = 256 / KP;             // vectors per DMA instruction (64 lanes x 16 B)
  static constexpr int NI = kStepNnz / VPI;        // DMA instructions per step
  static constexpr int LPV = 64 / VPI;             // lanes per vector
  static constexpr int VEC_BYTES = NI * 1024;      // one step of gathered vectors
  static constexpr int SLOT_BYTES = VEC_BYTES + 256 /* values */;
  static constexpr int IDX_SLOT = 256;
  static constexpr int WAVE_RING = kRing * SLOT_BYTES + kRing * IDX_SLOT;
  static constexpr int LDA = KP + 4;               // row stride of A: transposed 16-byte tile writes hit 8 bank groups
  // solve phase (aliases the rings): A | chain scratch | vectors
  static constexpr int A_BYTES = KP * LDA * 4;
  static constexpr int P_BYTES = NT * 16 * 64 * 4;
  static constexpr int V_FLOATS = 4 * KP /* b partials */ + 2 * 4 * KP /* matvec partials, double buffered */ +
                                  4 * KP /* per-wave published vector */ + 64 /* scalars */;
  static constexpr int SOLVE_BYTES = A_BYTES + P_BYTES + V_FLOATS * 4;
  static constexpr int RING_BYTES = 4 * WAVE_RING;
  static constexpr int BYTES = (SOLVE_BYTES > RING_BYTES ? SOLVE_BYTES : RING_BYTES) + 64;
  __host__ __device__ static constexpr int tile(int R, int C) { return R * (R + 1) / 2 + C; }
};

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(reinterpret_cast<uintptr_t>(p));  // low 32 bits of a generic LDS pointer = LDS byte address
}

// ---- LDS-DMA, issued from asm so that the loop's s_waitcnt can be counted (see the header) ----
// One 16-byte piece per lane: LDS destination = M0 + lane * 16 (wave-uniform base), source = each lane's own pointer.
__device__ __forceinline__ void dma16(const void* g, unsigned lds_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g), "s"(lds_base)
      : "memory");
}
// One dword per lane: LDS destination = M0 + lane * 4.
__device__ __forceinline__ void dma4(const void* g, unsigned lds_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g), "s"(lds_base)
      : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  const f32x2 f = {lo, hi};
  const bf16x2 h = __builtin_convertvector(f, bf16x2);  // v_cvt_pk_bf16_f32 (round to nearest even)
  return __builtin_bit_cast(unsigned, h);
}

// f[0..7] -> NS exact bf16 terms, packed as MFMA operands (element e of the operand = non-zero 8 * (lane / 32) + e)
template <int NS>
__device__ __forceinline__ void split8(const float (&f)[8], u32x4 (&parts)[NS]) {
#pragma unroll
  for (int j = 0; j < 4; j++) {
    float r0 = f[2 * j], r1 = f[2 * j + 1];
#pragma unroll
    for (int p = 0; p < NS; p++) {
      const unsigned pk = pack_bf16(r0, r1);
      parts[p][j] = pk;
      if (p + 1 < NS) {
        r0 -= __uint_as_float(pk << 16);
        r1 -= __uint_as_float(pk & 0xffff0000u);
      }
    }
  }
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

// Which wave accumulates what.  Rank 128 with implicit feedback needs 20 accumulator tiles (320 registers): more than
// one wave can hold next to its operands, so there the waves work in PAIRS on the same steps (two step sets instead of
// four) and split the tiles by role; everywhere else a wave owns every tile of its own steps.
//   NS = 3 (M1: 6 products per tile, M2: 3):  role 0 = M1 tile rows 2,3 (42 products), role 1 = M1 rows 0,1 + M2 (48)
//   NS = 2 (3 and 3):                         role 0 = M1, role 1 = M2
template <int KP, int NS, bool IMPLICIT>
struct NeRoles {
  static constexpr bool PAIR = IMPLICIT && KP == 128;
  static constexpr int NSETS = PAIR ? 2 : 4;
  // m: 0 = the matrix of the system (implicit M1, explicit M2), 1 = implicit M2 (loss only)
  __host__ __device__ static constexpr bool owns(int role, int m, int R, int /*C*/) {
    if (!PAIR) return true;
    if (NS >= 3) return m == 0 ? (role == 0 ? R >= 2 : R < 2) : role == 1;
    return m == role;
  }
  __host__ __device__ static constexpr bool owns_row(int role, int m, int R) { return owns(role, m, R, 0); }
  __host__ __device__ static constexpr int rhs_role() { return PAIR ? 1 : 0; }  // who accumulates b and sum c
};

// SOLVER: 1 = conjugate gradient on the assembled system, 0 = exact (Cholesky) solve in LDS
template <int KP, int NS, bool IMPLICIT, int SOLVER>
__global__ __launch_bounds__(256, 1) void als_ne_kernel(AlsArgs a, const int32_t* __restrict__ rows, int n_rows,
                                                         int* __restrict__ work_counter,
                                                         double* __restrict__ row_loss) {
  using G_ = NeGeo<KP>;
  using RL = NeRoles<KP, NS, IMPLICIT>;
  constexpr int NB = G_::NB, NT = G_::NT, NI = G_::NI, LPV = G_::LPV, LDA = G_::LDA;
  constexpr int NM = IMPLICIT ? 2 : 1;  // accumulated matrices: implicit {M1, M2}, explicit {M2}
  constexpr int NSETS = RL::NSETS;
  constexpr int LV = KP / 4;            // lanes of one copy of a rank-vector (4 floats per lane)
  constexpr int COPIES = 64 / LV;
  constexpr int RW = KP / 4;            // rows of A per wave in the matrix-vector product
  constexpr int RPC = RW / COPIES;      // ... per copy
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = rfl(tid >> 6);
  const int h = lane >> 5, d = lane & 31;
  const int k = a.k;
  const int wset = RL::PAIR ? wv >> 1 : wv;   // which steps of a row this wave works on: wset, wset + NSETS, ...
  const int wrole = RL::PAIR ? wv & 1 : 0;

  // accumulate phase: per-wave ring
  char* ring = smem + wv * G_::WAVE_RING;
  const unsigned ring_a = rfl((int)lds_addr(ring));
  auto vec_slot = [&](int i) { return ring_a + (unsigned)((i % kRing) * G_::SLOT_BYTES); };
  auto idx_slot = [&](int i) { return ring_a + (unsigned)(kRing * G_::SLOT_BYTES + (i % kRing) * G_::IDX_SLOT); };
  // solve phase
  float* sA = reinterpret_cast<float*>(smem);
  float* sP = reinterpret_cast<float*>(smem + G_::A_BYTES);
  float* sB = reinterpret_cast<float*>(smem + G_::A_BYTES + G_::P_BYTES);  // [4][KP]
  float* sRed = sB + 4 * KP;                                               // [2][4][KP]
  float* sPub = sRed + 2 * 4 * KP;                                         // [4][KP]
  float* sScal = sPub + 4 * KP;                                            // [64]
  int* sNext = reinterpret_cast<int*>(smem + G_::BYTES - 64);

  const int dq = lane / LPV;                 // which vector of a DMA instruction this lane copies
  const int dl4 = min((lane % LPV) * 4, k - 4);  // its 16-byte piece; pieces beyond the rank re-read the last one
                                                 // (never past the vector) and are zeroed when they are consumed
  int buf = 0;

  for (;;) {
    if (tid == 0) *sNext = atomicAdd(work_counter, 1);
    __syncthreads();
    const int ri = rfl(*sNext);
    if (ri >= n_rows) break;
    const int row = rfl(rows[ri]);
    const int p1 = rfl(a.col_ptrs[row]), p2 = rfl(a.col_ptrs[row + 1]);
    const int cnt = p2 - p1;
    const int nsteps = (cnt + kStepNnz - 1) / kStepNnz;
    const int nst = (nsteps - wset + NSETS - 1) / NSETS;   // this wave's steps: global steps wset, wset + NSETS, ...
    float* yrow = a.Y + (size_t)row * k;
    const float lam_use = IMPLICIT ? 0.f : (float)(a.lambda_loss * (a.dynamic_lambda ? (double)(float)cnt : 1.0));

    // ---- gather pipeline (no selects: every source address is valid, padding is zeroed at consumption) ----
    auto issue_idx = [&](int i) {  // index chunk of this wave's step i (and the 48 entries behind it), clamped to the row
      const int pos = min(p1 + (wset + NSETS * i) * kStepNnz + lane, p2 - 1);
      dma4(a.row_idx + pos, idx_slot(i));
    };
    auto issue_vec = [&](int i) {  // needs the index chunk of step i in LDS
      const int s0 = (wset + NSETS * i) * kStepNnz;
      const int* ix = reinterpret_cast<const int*>(smem + (idx_slot(i) - lds_addr(smem))) + dq * NI;
      int id[NI];
      if constexpr (NI == 8) {
        const int4 i0 = *reinterpret_cast<const int4*>(ix), i1 = *reinterpret_cast<const int4*>(ix + 4);
        id[0] = i0.x; id[1] = i0.y; id[2] = i0.z; id[3] = i0.w;
        id[4] = i1.x; id[5] = i1.y; id[6] = i1.z; id[7] = i1.w;
      } else {
        const int4 i0 = *reinterpret_cast<const int4*>(ix);
        id[0] = i0.x; id[1] = i0.y; id[2] = i0.z; id[3] = i0.w;
      }
      const unsigned base = vec_slot(i);
#pragma unroll
      for (int e = 0; e < NI; e++) dma16(a.X + (size_t)id[e] * k + dl4, base + e * 1024);
      const int pos = min(p1 + s0 + lane, p2 - 1);
      dma4(a.vals + pos, base + G_::VEC_BYTES);
    };
    constexpr int GROUP = NI + 2;  // DMA instructions per pipeline group: 1 index chunk + NI vector pieces + 1 value chunk

    auto body = [&](auto role_tag) {
      constexpr int ROLE = decltype(role_tag)::value;
      f32x16 acc[NM][NT];
#pragma unroll
      for (int m = 0; m < NM; m++)
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
          for (int e = 0; e < 16; e++) acc[m][t][e] = 0.f;
      float bp[NB], bp_hi[NB];
#pragma unroll
      for (int t = 0; t < NB; t++) bp[t] = bp_hi[t] = 0.f;
      double sc = 0.0;
      constexpr bool RHS = ROLE == RL::rhs_role();

      if (nst > 0) {
        issue_idx(0);
        issue_idx(1);
        wait_vm<0>();
        issue_idx(2);
        issue_vec(0);
        issue_idx(3);
        if (nst > 1) issue_vec(1);
      }
      // one pipeline step; MASKED = the row's last (partial) step, or rank < KP
      auto iter = [&](auto masked_tag, const int i) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        // group i-2 (index chunk i+2, vectors i) must have landed; group i-1 may stay in flight
        if (i + 1 < nst) wait_vm<GROUP>(); else wait_vm<1>();
        issue_idx(i + 4);
        if (i + 2 < nst) issue_vec(i + 2);

        const int s0 = (wset + NSETS * i) * kStepNnz;
        const int rem = cnt - s0;
        const char* slot = smem + (vec_slot(i) - lds_addr(smem));
        float c[8];  // values of this lane's 8 non-zeros (8h .. 8h+7)
        {
          const float4 c0 = *reinterpret_cast<const float4*>(slot + G_::VEC_BYTES + h * 32);
          const float4 c1 = *reinterpret_cast<const float4*>(slot + G_::VEC_BYTES + h * 32 + 16);
          c[0] = c0.x; c[1] = c0.y; c[2] = c0.z; c[3] = c0.w;
          c[4] = c1.x; c[5] = c1.y; c[6] = c1.z; c[7] = c1.w;
        }
        if constexpr (MASKED) {
#pragma unroll
          for (int e = 0; e < 8; e++) c[e] = (8 * h + e < rem) ? c[e] : 0.f;
        }
        if constexpr (RHS) {
          float s = 0.f;
#pragma unroll
          for (int e = 0; e < 8; e++) s += IMPLICIT ? c[e] : c[e] * c[e];
          sc += (double)s;
        }
        // operands: lane (h, d) holds, for block t, dimension 32 t + d of non-zeros 8h .. 8h+7
        const char* lbase = slot + h * (8 / NI) * (KP * 4) + d * 4;
        auto load_block = [&](int t, float (&raw)[8]) {
#pragma unroll
          for (int e = 0; e < 8; e++)
            raw[e] = *reinterpret_cast<const float*>(lbase + (e % NI) * 1024 + (e / NI) * (KP * 4) + t * 128);
          if constexpr (MASKED) {  // padding slots of a row's last step hold copies of its last vector; dimensions
            const bool live = 32 * t + d < k;  // beyond the rank hold copies of the vector's last piece
#pragma unroll
            for (int e = 0; e < 8; e++) raw[e] = (live && 8 * h + e < rem) ? raw[e] : 0.f;
          }
        };
        u32x4 xp[NB][NS];
#pragma unroll
        for (int t = 0; t < NB; t++) {
          float raw[8];
          load_block(t, raw);
          if constexpr (RHS) {
#pragma unroll
            for (int e = 0; e < 8; e++) bp[t] = fmaf(c[e], raw[e], bp[t]);
          }
          split8<NS>(raw, xp[t]);
        }
        if constexpr (!IMPLICIT) {  // M2 = X X^T at full split precision
#pragma unroll
          for (int pa = 0; pa < NS; pa++)
#pragma unroll
            for (int pb = 0; pb + pa < NS; pb++)
#pragma unroll
              for (int R = 0; R < NB; R++)
#pragma unroll
                for (int C = 0; C <= R; C++)
                  acc[0][G_::tile(R, C)] = mfma_bf16(xp[R][pa], xp[C][pb], acc[0][G_::tile(R, C)]);
        } else {  // M2 from the first two terms (loss only), M1 = (W X) X^T at full split precision
#pragma unroll
          for (int pa = 0; pa < 2; pa++)
#pragma unroll
            for (int pb = 0; pb + pa < 2; pb++)
#pragma unroll
              for (int R = 0; R < NB; R++)
#pragma unroll
                for (int C = 0; C <= R; C++)
                  if constexpr (RL::owns(ROLE, 1, 0, 0))
                    acc[1][G_::tile(R, C)] = mfma_bf16(xp[R][pa], xp[C][pb], acc[1][G_::tile(R, C)]);
#pragma unroll
          for (int R = 0; R < NB; R++) {
            if (RL::owns_row(ROLE, 0, R)) {
              float ar[8];  // a = (c - 1) * x for block R, re-read from the ring (cheaper than holding the raw values)
              load_block(R, ar);
#pragma unroll
              for (int e = 0; e < 8; e++) ar[e] *= c[e] - 1.f;
              u32x4 ap[NS];
              split8<NS>(ar, ap);
#pragma unroll
              for (int pa = 0; pa < NS; pa++)
#pragma unroll
                for (int pb = 0; pb + pa < NS; pb++)
#pragma unroll
                  for (int C = 0; C <= R; C++)
                    acc[0][G_::tile(R, C)] = mfma_bf16(ap[pa], xp[C][pb], acc[0][G_::tile(R, C)]);
            }
          }
        }
        if (RHS && (i & 63) == 63) {  // two-level sum of the right-hand side: bounds the fp32 running-sum error
#pragma unroll
          for (int t = 0; t < NB; t++) {
            bp_hi[t] += bp[t];
            bp[t] = 0.f;
          }
        }
      };
      if (k == KP) {
        // only the row's last step can be partial, and it belongs to exactly one wave set
        const int nfull = (nst > 0 && cnt - (wset + NSETS * (nst - 1)) * kStepNnz < kStepNnz) ? nst - 1 : nst;
        for (int i = 0; i < nfull; i++) iter(std::false_type{}, i);
        if (nfull < nst) iter(std::true_type{}, nfull);
      } else {
        for (int i = 0; i < nst; i++) iter(std::true_type{}, i);
      }
      wait_vm<0>();  // trailing index chunks
#pragma unroll
      for (int t = 0; t < NB; t++) bp[t] = half_swap_sum(bp_hi[t] + bp[t]);
      sc += __shfl_xor(sc, 32);  // sc is uniform inside each half of the wave
      __syncthreads();  // every wave is done with its ring: the solve phase may overwrite it

      // ---- reduce the partial systems in a fixed order (step set 0, 1, ...: deterministic) through a tile-major scratch
      if (lane < 32) {
#pragma unroll
        for (int t = 0; t < NB; t++) sB[wv * KP + 32 * t + lane] = RHS ? bp[t] : 0.f;
      }
      if (lane == 0) reinterpret_cast<double*>(sScal)[wv] = RHS ? sc : 0.0;
      // warm start and the Gramian entries of the tiles this wave will finish (requested now, used after the chain)
      float x[4];
      {
        const int off = (lane % LV) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (SOLVER == 1 && off < k) v = *reinterpret_cast<const float4*>(yrow + off);
        x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
      }
      constexpr int TPW = (NT + 3) / 4;  // tiles finished per wave
      float gt[TPW][16];
#pragma unroll
      for (int u = 0; u < TPW; u++) {
        const int t = wv + 4 * u;
        int R = 0;
        while ((R + 1) * (R + 2) / 2 <= t) R++;
        const int C = t - R * (R + 1) / 2;
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int i = 32 * R + 8 * (e >> 2) + 4 * h + (e & 3), j = 32 * C + d;
          float g = 0.f;
          if (t < NT && i < k && j < k) {
            if constexpr (IMPLICIT) g = a.XtX[(size_t)i * k + j];
            else g = i == j ? lam_use : 0.f;
          }
          gt[u][e] = g;
        }
      }
      for (int ph = 0; ph < NSETS; ph++) {
        if (wset == ph) {
#pragma unroll
          for (int R = 0; R < NB; R++)
#pragma unroll
            for (int C = 0; C <= R; C++)
              if (RL::owns(ROLE, 0, R, 0)) {
                const int t = G_::tile(R, C);
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++) {
                  float4* slot4 = reinterpret_cast<float4*>(sP) + (t * 4 + q4) * 64 + lane;
                  float4 v = make_float4(acc[0][t][4 * q4], acc[0][t][4 * q4 + 1], acc[0][t][4 * q4 + 2],
                                         acc[0][t][4 * q4 + 3]);
                  if (ph > 0) {
                    const float4 o = *slot4;
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                  }
                  *slot4 = v;
                }
              }
        }
        __syncthreads();
      }
      // A = XtX + M1 (explicit: M2 + lambda_use I), full symmetric, row stride LDA
#pragma unroll
      for (int u = 0; u < TPW; u++) {
        const int t = wv + 4 * u;
        if (t < NT) {
          int R = 0;
          while ((R + 1) * (R + 2) / 2 <= t) R++;
          const int C = t - R * (R + 1) / 2;
#pragma unroll
          for (int q4 = 0; q4 < 4; q4++) {
            const float4 o = reinterpret_cast<const float4*>(sP)[(t * 4 + q4) * 64 + lane];
            const float v[4] = {o.x + gt[u][4 * q4], o.y + gt[u][4 * q4 + 1], o.z + gt[u][4 * q4 + 2],
                                o.w + gt[u][4 * q4 + 3]};
            const int i0 = 32 * R + 8 * q4 + 4 * h, j = 32 * C + d;
#pragma unroll
            for (int rr = 0; rr < 4; rr++) sA[(i0 + rr) * LDA + j] = v[rr];
            if (R != C) *reinterpret_cast<float4*>(sA + j * LDA + i0) = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
      }
      __syncthreads();

      // ---- right-hand side, replicated in every wave: lane holds elements [4 (lane % LV), +4)
      const int c4 = (lane % LV) * 4, copy = lane / LV;
      float b[4];
      {
        float4 s = *reinterpret_cast<const float4*>(sB + c4);
#pragma unroll
        for (int w2 = 1; w2 < 4; w2++) {
          const float4 o = *reinterpret_cast<const float4*>(sB + w2 * KP + c4);
          s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
        }
        b[0] = s.x; b[1] = s.y; b[2] = s.z; b[3] = s.w;
      }
      const double sc_row = (reinterpret_cast<const double*>(sScal)[0] + reinterpret_cast<const double*>(sScal)[1]) +
                            (reinterpret_cast<const double*>(sScal)[2] + reinterpret_cast<const double*>(sScal)[3]);

      // out = A v: every wave takes RW rows of A, its COPIES lane groups RPC rows each; partial vectors through LDS
      float* pub = sPub + wv * KP;
      auto matvec = [&](const float (&v)[4], float (&out)[4]) {
        wave_sync();
        if (copy == 0) *reinterpret_cast<float4*>(pub + c4) = make_float4(v[0], v[1], v[2], v[3]);
        wave_sync();
        const int r0 = wv * RW + copy * RPC;
        float o4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q4 = 0; q4 < RPC / 4; q4++) {
          const float4 vb = *reinterpret_cast<const float4*>(pub + r0 + 4 * q4);
          const float vv[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const float4 ar = *reinterpret_cast<const float4*>(sA + (r0 + 4 * q4 + u) * LDA + c4);
            o4[0] = fmaf(vv[u], ar.x, o4[0]);
            o4[1] = fmaf(vv[u], ar.y, o4[1]);
            o4[2] = fmaf(vv[u], ar.z, o4[2]);
            o4[3] = fmaf(vv[u], ar.w, o4[3]);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) o4[u] = COPIES == 2 ? half_swap_sum(o4[u]) : groups_sum_ne(o4[u]);
        float* red = sRed + buf * 4 * KP;
        if (copy == 0) *reinterpret_cast<float4*>(red + wv * KP + c4) = make_float4(o4[0], o4[1], o4[2], o4[3]);
        __syncthreads();
        float4 s = *reinterpret_cast<const float4*>(red + c4);
#pragma unroll
        for (int w2 = 1; w2 < 4; w2++) {
          const float4 o = *reinterpret_cast<const float4*>(red + w2 * KP + c4);
          s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
        }
        out[0] = s.x; out[1] = s.y; out[2] = s.z; out[3] = s.w;
        buf ^= 1;
      };
      auto dot = [&](const float (&u)[4], const float (&v)[4]) {
        float s = u[0] * v[0];
        s = fmaf(u[1], v[1], s);
        s = fmaf(u[2], v[2], s);
        s = fmaf(u[3], v[3], s);
        return wave_sum_all(s) * (1.f / COPIES);  // the COPIES lane groups hold identical values
      };

      float r[4], p[4], ap[4];
      if constexpr (SOLVER == 1) {
        // cg_solver_implicit / cg_solver_explicit on the assembled operator; rsold / alpha in double like the
        // reference (wrmf_implicit.hpp:18)
        matvec(x, ap);
#pragma unroll
        for (int u = 0; u < 4; u++) p[u] = r[u] = b[u] - ap[u];
        double rsold = (double)dot(r, r);
        bool conv = false;
        for (int it = 0; it < a.cg_steps; ++it) {
          matvec(p, ap);  // uniform control flow: every wave runs the barrier inside
          if (!conv) {
            const float alpha = (float)(rsold / (double)dot(p, ap));
#pragma unroll
            for (int u = 0; u < 4; u++) {
              x[u] = fmaf(alpha, p[u], x[u]);
              r[u] = fmaf(-alpha, ap[u], r[u]);
            }
            const double rsnew = (double)dot(r, r);
            if (rsnew < (double)kCgTolNe) {
              conv = true;
            } else {
              const float beta = (float)(rsnew / rsold);
#pragma unroll
              for (int u = 0; u < 4; u++) p[u] = fmaf(p[u], beta, r[u]);
              rsold = rsnew;
            }
          }
        }
      }

      // ---- loss of the row: sum c - 2 y.b + y^T (M1 + M2) y + lambda |y|^2  (explicit: sum r^2 - 2 y.b + y^T M2 y)
      wave_sync();
      if (copy == 0) *reinterpret_cast<float4*>(pub + c4) = make_float4(x[0], x[1], x[2], x[3]);
      wave_sync();
      float qf = 0.f;
      {
        float yj[NB];
#pragma unroll
        for (int t = 0; t < NB; t++) yj[t] = pub[32 * t + d];
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
              if (RL::owns(ROLE, m, R, 0)) {
#pragma unroll
                for (int e = 0; e < 16; e++) s = fmaf(acc[m][G_::tile(R, C)][e], yi[e], s);
              }
            qf = fmaf(R == C ? 1.f : 2.f, s * yj[C], qf);
          }
        }
        qf = wave_sum_all(qf);
      }
      if (lane == 0) sScal[16 + wv] = qf;
      const float yb = dot(x, b), yy = dot(x, x);
      __syncthreads();
      if (wv == 0) {
        const double q = ((double)sScal[16] + (double)sScal[17]) + ((double)sScal[18] + (double)sScal[19]);
        const double fit = sc_row - 2.0 * (double)yb + q;
        const double reg = IMPLICIT ? a.lambda_loss * (double)yy : (double)(lam_use * yy);
        if (lane == 0) row_loss[ri] = fit + reg;
        if (copy == 0 && c4 < k) *reinterpret_cast<float4*>(yrow + c4) = make_float4(x[0], x[1], x[2], x[3]);
      }
      __syncthreads();  // the next row's rings overwrite the solve area
    };
    if (!RL::PAIR || wrole == 0) body(std::integral_constant<int, 0>{});
    else body(std::integral_constant<int, 1>{});
  }
}

template <int KP, int NS, bool IMPLICIT, int SOLVER>
hipError_t launch_ne_t(const AlsArgs& a, const int32_t* rows, int n_rows, int* counter, double* row_loss, int grid,
                       hipStream_t s) {
  auto kern = als_ne_kernel<KP, NS, IMPLICIT, SOLVER>;
  constexpr int lds = NeGeo<KP>::BYTES;
  hipError_t err =
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (err != hipSuccess) return err;
  if ((err = hipMemsetAsync(counter, 0, sizeof(int), s)) != hipSuccess) return err;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a, rows, n_rows, counter, row_loss);
  return hipGetLastError();
}

}  // namespace

bool ne_supported(int k) { return k > 32 && k <= 128 && k % 4 == 0; }

// rows[0, n_rows): schedule order (longest first); row_loss: one double per row; counter: one int of scratch
hipError_t launch_als_ne(const AlsArgs& a, const int32_t* rows, int n_rows, bool implicit, int* counter,
                         double* row_loss, hipStream_t s) {
  if (n_rows <= 0) return hipSuccess;
  int dev = 0, cus = 256;
  hipError_t err = hipGetDevice(&dev);
  if (err != hipSuccess) return err;
  if ((err = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return err;
  const int grid = n_rows < cus ? n_rows : cus;  // one workgroup per CU (the kernel owns the CU's register file)
  const int KP = padded_rank(a.k);
#ifndef RSP_NE_SPLIT
#define RSP_NE_SPLIT 3
#endif
  if (KP == 128)
    return implicit ? launch_ne_t<128, RSP_NE_SPLIT, true, 1>(a, rows, n_rows, counter, row_loss, grid, s)
                    : launch_ne_t<128, RSP_NE_SPLIT, false, 1>(a, rows, n_rows, counter, row_loss, grid, s);
  if (KP == 64)
    return implicit ? launch_ne_t<64, RSP_NE_SPLIT, true, 1>(a, rows, n_rows, counter, row_loss, grid, s)
                    : launch_ne_t<64, RSP_NE_SPLIT, false, 1>(a, rows, n_rows, counter, row_loss, grid, s);
  return hipErrorInvalidValue;
}

}  // namespace rsparse_hip
