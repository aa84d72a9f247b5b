// Exact (Cholesky) half-iteration at rank 65..128: one WAVE per row, the row's 128 x 128 system in the wave's matrix-core
// ACCUMULATOR registers from the assembly to the factorisation (gfx950, wave64; round 6).
//
// Replaces, for the rows of 65 .. kCholMfMax non-zeros, the exact-solve instantiation of wrmf_ne.hip -- the
// solver == CHOLESKY branch of als_implicit<T> / als_explicit<T> (inst/include/wrmf_implicit.hpp:207-208,231,236;
// wrmf_explicit.hpp:103-108):
//     lhs = XtX + X_nnz diag(c - 1) X_nnz^T   |   X_nnz X_nnz^T + lambda_use I,      rhs = X_nnz c,
//     Y_new = solve(lhs, rhs, fast + likely_sympd).
// That kernel assembles a row with four waves and then factors it with the same four (wrmf_ldlt.h): 66 k cycles per row for
// 6 k cycles of FMAs -- a chain of 128 pivots handed from wave to wave, two workgroups per CU.  Here nothing
// is shared and nothing is handed over:
//   * assembly: as als_chol_wave_kernel (rank <= 64) -- the lane = coordinate layout in which the vectors arrive is the
//     operand layout of v_mfma_f32_32x32x16_f16 once the halves of the wave have traded registers; operands as two fp16 terms
//     of 2^e x (and 2^e' (c - 1) x), three products of order < 2 (2^-21 per product, fp32 accumulation).  The TEN lower 32 x 32
//     tiles of the symmetric system are 160 accumulator registers: tile (I, K), I >= K, holds at lane (n, hf), register v the
//     entry [row 32 I + n][column 32 K + rho(v, hf)], rho = 8 (v >> 2) + 4 hf + (v & 3) -- "lane = row, register = column";
//   * factorisation: right-looking blocked Cholesky, panels of 8 columns.  In this layout the 8 columns of panel q of block
//     column K are registers 4 q .. 4 q + 3 of the tiles (I, K): lanes (n, 0) hold columns 0..3 of row n, lanes (n, 1)
//     columns 4..7.  Per panel: the 8 x 8 diagonal block goes through 256 bytes of LDS to every lane and is factored there
//     redundantly (the same bits in each lane: no exchange); every lane solves ITS row against it (the half that holds columns
//     4..7 gets the other half's results by one lane swap per register); and the trailing update is
//         tile (I2, K2) -= P_{K2} P_{I2}^T        4 x v_mfma_f32_32x32x2_f32 (exact fp32 products)
//     whose A and B operands ARE the panel's registers as they stand: register m of a tile is column m at the lanes (n, 0)
//     and column 4 + m at the lanes (n, 1), which is the k = 2 operand layout (a contraction does not care about the order
//     of its terms).  No transposition, no LDS, no barrier; rows above the panel are masked to zero in the operand, which also
//     keeps the finished columns of a tile untouched.  The forward substitution rides along (right-hand side per lane = row,
//     the two halves of the wave carrying partial sums), the backward substitution walks the panels in reverse (a 32-lane
//     reduction per column);
//   * the 16 panel steps are a LOOP whose body moves the panel's registers to fixed temporaries in a switch: 10 KB of code
//     instead of 70 KB of unrolled steps for eight waves per CU in different phases (the instruction cache is 64 KB per two CUs).
//   * the 160 accumulator registers are the ACCUMULATOR file (a0 .. a159) by name, through inline asm: as ten f32x16 values
//     of the compiler's they were copied tile-wise at every join of the switch and two to six tiles lived in scratch (first
//     version: 338 ms per launch for rows that the kernel it replaces solved in 128).  The compiler now sees a kernel of
//     <= 96 vector registers (amdgpu_num_vgpr) and two waves per SIMD fit: 96 + 160 = 256.  hipcc pads no hazard inside an
//     asm statement: every matrix instruction opens with s_nop 1 (a VALU-written operand), every read of the accumulator
//     file sits behind an explicit s_nop block (a 16-pass result), see MF_DRAIN.
// A non-positive pivot sends the row to the general solver (wrmf_lu.hip), exactly as the other exact kernels do.
#include <cstdio>
#include <utility>

#pragma clang diagnostic ignored "-Winline-asm"   // (the named accumulator registers are "reserved": that is the point)

#include "wrmf_internal.h"
#include "wrmf_device.h"

namespace rsparse_hip {
namespace {

using namespace dev;

// Who owns which register.  hipcc gives a kernel that uses the accumulator file HALF of its vector-register budget there and
// spills into it; LLVM's function attribute "amdgpu-agpr-alloc"="0" takes that away (the compiler then never allocates an
// accumulator register and its budget -- twice amdgpu_num_vgpr -- is all vector registers: 96), but the attribute has no source
// spelling.  rsparse_amd/build.py compiles THIS file with -mllvm -forceattrs-csv-path=wrmf_chol_mf.attrs.csv (LLVM's
// ForceFunctionAttrs pass; the csv names the three kernels below, which is why they are extern "C") and then audits the
// listing: no compiler-generated accumulator-file instruction, accum_offset 96, 160 accumulator registers -- 96 + 160 = 256, two
// waves per SIMD.  -DMF_SAFE (what a build without the csv must use; build.py falls back to it when the audit fails): the
// compiler keeps its half, the tiles sit above it, a[96 : 255], one wave per SIMD.
#ifdef MF_SAFE
constexpr int MF_A0 = 96;
#define MF_TOP "a255"
#define MF_NUM_VGPR 96
#define MF_WAVES 1
#else
constexpr int MF_A0 = 0;
#define MF_TOP "a159"
#define MF_NUM_VGPR 48
#define MF_WAVES 2
#endif

}  // namespace
}  // namespace rsparse_hip

#include "wrmf_mf.h"

namespace rsparse_hip {
namespace {

template <int NOPS>
struct MfSmem {
  float D[64];      // the diagonal block of the current panel, row-major 8 x 8
  float Bp[16];     // the right-hand side of its rows: [hf][row], the two halves' partial sums
  float U[128];     // rhs, then u = L^-1 rhs, then y
  float B[2][128];  // the right-hand side per row while the panels run: the two halves' partial sums
  float Inv[128];   // 1 / L_cc
  float L8[16][64]; // the factored diagonal blocks (lower triangles), for the backward pass
  float4 ops[NOPS][64]; // the assembly loop's operands between their split and their products (see there)
  double loss;      // this wave's sum of row terms (kept out of the registers: the kernel runs at exactly 96)
};

// panel (K, Q): registers 4 Q .. 4 Q + 3 of the tiles (K + s, K), s < 4 - K; the slots [S0, S1) of them
template <int K, int Q, int S0, int S1>
__device__ __forceinline__ void mf_copy_out(float (&P)[4][4]) {
  mf_sfor<(S1 < 4 - K ? S1 : 4 - K) - S0>([&](auto st) {
    constexpr int s = S0 + decltype(st)::value;
    mf_sfor<4>([&](auto mt) {
      constexpr int m = decltype(mt)::value;
      P[s][m] = mf_rd<16 * mf_tid(K + s, K) + 4 * Q + m>();
    });
  });
}

template <int K, int Q>
__device__ __forceinline__ void mf_copy_in_update(const float (&Pm)[4][4]) {
  // (the rows at and above the panel's diagonal block are stored as the zeros the operands need: nothing reads them again --
  // the backward pass takes the diagonal blocks from LDS)
  mf_sfor<4 - K>([&](auto st) {
    constexpr int s = decltype(st)::value;
    mf_sfor<4>([&](auto mt) {
      constexpr int m = decltype(mt)::value;
      mf_wr<16 * mf_tid(K + s, K) + 4 * Q + m>(Pm[s][m]);
    });
  });
  // tile (K + s, K + s2) -= P_{s2} P_s^T: register m = columns (m, 4 + m).  m outermost: consecutive matrix instructions go
  // to different tiles (a tile's four are a dependent chain).  The tiles of block column K are skipped after its last panel
  // (the operand is all zeros there).
  mf_sfor<4>([&](auto mt) {
    constexpr int m = decltype(mt)::value;
    mf_sfor<4 - K>([&](auto k2t) {
      constexpr int s2 = decltype(k2t)::value;   // K2 = K + s2
      if constexpr (s2 > 0 || Q < 3) {
        const float na = -Pm[s2][m];
        mf_sfor<4 - K - s2>([&](auto it) {
          constexpr int s = s2 + decltype(it)::value;  // I2 = K + s >= K2
          mf_mma2<mf_tid(K + s, K + s2)>(na, Pm[s][m]);
        });
      }
    });
  });
}

#define MF_SWITCH16(p, CALL)                                                                                  \
  switch (p) {                                                                                                \
    case 0: CALL(0, 0); break;   case 1: CALL(0, 1); break;   case 2: CALL(0, 2); break;   case 3: CALL(0, 3); break;   \
    case 4: CALL(1, 0); break;   case 5: CALL(1, 1); break;   case 6: CALL(1, 2); break;   case 7: CALL(1, 3); break;   \
    case 8: CALL(2, 0); break;   case 9: CALL(2, 1); break;   case 10: CALL(2, 2); break;  case 11: CALL(2, 3); break;  \
    case 12: CALL(3, 0); break;  case 13: CALL(3, 1); break;  case 14: CALL(3, 2); break;  default: CALL(3, 3); break;  \
  }

template <bool IMPLICIT, bool SYM, bool KFULL>
__device__ __forceinline__ void als_chol_mf_body(const AlsArgs& a, const int32_t* __restrict__ rows, const int n_rows, const int loss_slot0) {
  __shared__ __attribute__((aligned(16))) MfSmem<SYM ? 8 : 16> sm;
  const int lane = threadIdx.x & 63;
  const int k = a.k;
  const bool vec = (k % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.X) & 15) == 0);
  if constexpr (IMPLICIT) {
    // two instantiations are launched; "some confidence < 1" (word 2 of the values scan) says which one works, the other
    // leaves at once (its loss slots zero)
    const bool below_one = a.wave_stats[2] != 0u;
    if (below_one == SYM) {
      if (lane == 0) a.loss_partials[loss_slot0 + blockIdx.x] = 0.0;
      return;
    }
  }

  if (lane == 0) sm.loss = 0.0;
#ifdef RSP_MF_PROF   // dev builds (tools/gpu_mf_prof.sh): s_memtime ticks per phase, summed over the waves into a.ne_prof[8 ..]
  unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt0 = __builtin_amdgcn_s_memtime();
#define MF_TICK(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pt[i] += t_ - pt0; pt0 = t_; }
#else
#define MF_TICK(i)
#endif
  // operand scales: powers of two from max |X| and max c (launch_ne_stats), as in wrmf_chol_wave.hip.  (Through
  // readfirstlane: loaded values are vector registers to hipcc, and a dozen uniform constants held in vector registers
  // through every phase are a dozen registers the assembly loop then spills its operands for.)
  auto uni = [](const float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); };
  const int ex = rfl(mf_scale_exp(fmaxf(__uint_as_float(a.wave_stats[0]), 1e-30f)));
  const float wmax = uni(IMPLICIT ? fmaxf(__uint_as_float(a.wave_stats[1]) - 1.f, 1.f) : 1.f);
  const int ewb = rfl((int)((__float_as_uint(wmax) >> 23) & 0xffu));
  const float sx = uni(mf_pow2(ex)), sw = uni(mf_pow2(min(253, max(1, 253 - ewb))));   // sw = 2^(126 - ewb) <= 1 / wmax
  const float un1 = uni(mf_pow2(254 - ex));                                             // 1 / sx
  const float unw = uni(IMPLICIT ? mf_pow2(254 - min(253, max(1, 253 - ewb))) : 1.f);   // 1 / sw

  for (int it = blockIdx.x; it < n_rows; it += gridDim.x) {
    const int row = rfl(rows[it]);
    const int p1 = rfl(a.col_ptrs[row]), p2 = rfl(a.col_ptrs[row + 1]);
    float* yrow = a.Y + (size_t)row * k;
    const float lam_use = IMPLICIT ? 0.f : uni((float)(a.lambda_loss * (a.dynamic_lambda ? (double)(float)(p2 - p1) : 1.0)));
    // the lane id as this row sees it (keeps lane-dependent addresses and compares out of the registers across rows)
    // (re-derived per PHASE: what a later phase makes of the lane id would otherwise sit in registers through the assembly loop,
    // which then spilled operand registers and reloaded them behind `s_waitcnt vmcnt(0)` -- in front of the products that are
    // meant to run while the gather flies)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int lc0 = min(ln, k - 1), lc1 = min(ln + 64, k - 1);

    // ---------------- assembly on the matrix cores ----------------
    MF_TICK(7)
    MF_DRAIN();   // (the previous row's last matrix instructions)
    mf_sfor<160>([&](auto rt) { mf_wr<decltype(rt)::value>(0.f); });
    float u0 = 0.f, u1 = 0.f;   // rhs, lane = coordinate (ln, 64 + ln)
    MF_TICK(0)
    {
      // A step = 16 non-zeros.  Its vectors are requested at the top of the loop body (asm loads: one coordinate per lane
      // and register, no wait), the matrix instructions of the PREVIOUS step run while they fly, then ONE wait, the
      // right-hand side, the scaling and the split into fp16 terms.  Nothing in flight crosses the loop's back edge and
      // the body needs 32 (in flight) + 32 (operands) + ~16 registers: hipcc has no reason to copy or spill a register whose
      // data has not landed (it does not know: an asm load's destination counts as written at once) -- build.py's audit
      // demands a listing without a single scratch instruction.
      const int vo0 = 4 * lc0, vo1 = 4 * lc1;
      const float m0 = ln < k ? 1.f : 0.f, m1 = ln + 64 < k ? 1.f : 0.f;
      // The operands of a step wait for their products (which run behind the NEXT step's requests) in LDS, not in registers:
      // as eight register tuples carried around the loop hipcc kept six of them in scratch and reloaded them behind
      // `s_waitcnt vmcnt(0)` -- i.e. behind the gather the products are meant to overlap (20 registers stayed unused).
      // 16 KB of LDS traffic per step and wave, counted by lgkmcnt.
      auto stash = [&](const int t, const f16x8& v) { sm.ops[t][ln] = __builtin_bit_cast(float4, v); };
      auto fetch_op = [&](const int t) { return __builtin_bit_cast(f16x8, sm.ops[t][ln]); };
      auto products = [&]() {
        f16x8 bh[4], bl[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
          bh[t] = fetch_op(t);
          bl[t] = fetch_op(4 + t);
        }
        if constexpr (SYM) {
          // three products of order < 2, product-outermost: consecutive instructions go to different tiles
          mf_sfor<3>([&](auto pt) {
            constexpr int pr = decltype(pt)::value;
            mf_sfor<4>([&](auto kt) {
              constexpr int K = decltype(kt)::value;
              mf_sfor<4 - K>([&](auto st) {
                constexpr int I = K + decltype(st)::value;
                mf_mma16<mf_tid(I, K)>(pr == 2 ? bl[K] : bh[K], pr == 1 ? bl[I] : bh[I]);
              });
            });
          });
        } else {
          // the A side (2^e' (c - 1) 2^e x) one block column at a time
          mf_sfor<4>([&](auto kt) {
            constexpr int K = decltype(kt)::value;
            const f16x8 ah = fetch_op(8 + K), al = fetch_op(12 + K);
            mf_sfor<3>([&](auto pt) {
              constexpr int pr = decltype(pt)::value;
              mf_sfor<4 - K>([&](auto st) {
                constexpr int I = K + decltype(st)::value;
                mf_mma16<mf_tid(I, K)>(pr == 2 ? al : ah, pr == 1 ? bl[I] : bh[I]);
              });
            });
          });
        }
      };
      // lane j (mod 16) holds non-zero j of a step; the slots beyond the row repeat its last non-zero with weight 0.  Index and
      // value are requested a step ahead (two registers) and first looked at when the step's other work is done: one memory
      // round trip per step in front of the vector requests instead of two
      int idn = a.row_idx[p1 + min(ln & 15, min(16, p2 - p1) - 1)];
      float cvn = a.vals[p1 + min(ln & 15, min(16, p2 - p1) - 1)];
      for (int base = p1; base < p2; base += 16) {
        const int ccnt = min(16, p2 - base);
        const int idj = idn;
        const float cvr = cvn;
        {
          const int nb = min(base + 16, p2 - 1);   // (the last step asks for an entry it has: no branch)
          const int jn = nb + min(ln & 15, max(min(16, p2 - nb), 1) - 1);
          idn = a.row_idx[jn];
          cvn = a.vals[jn];
        }
        const bool inl = (ln & 15) < ccnt;
        const float cvj = inl ? cvr : 0.f;
        float fj;   // the scale of the slot's operand
        if constexpr (SYM && IMPLICIT) fj = inl ? sx * __builtin_amdgcn_sqrtf(fmaxf(cvr - 1.f, 0.f) * sw) : 0.f;
        else fj = inl ? sx : 0.f;
        const float wj = (cvr - 1.f) * sw;   // (!SYM: the A side's extra factor)
        float xs0[16], xs1[16];
#pragma unroll
        for (int s2 = 0; s2 < 16; s2++) {
          const float* bp = a.X + (size_t)__builtin_amdgcn_readlane(idj, s2) * k;
          mf_ld(xs0[s2], bp, vo0);
          mf_ld(xs1[s2], bp, vo1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (base > p1) products();   // (wave-uniform) the previous step's, behind this step's requests
        __builtin_amdgcn_sched_barrier(0);
        mf_wait(xs0, xs1);
#pragma unroll
        for (int s2 = 0; s2 < 16; s2++) {
          const float cv = readlane_f(cvj, s2);
          u0 = fmaf(cv, xs0[s2], u0);   // (lanes beyond the rank: masked when u is read)
          u1 = fmaf(cv, xs1[s2], u1);
          const float sc = readlane_f(fj, s2);
          xs0[s2] *= sc;
          xs1[s2] *= sc;
        }
        asm volatile("" : "+v"(u0), "+v"(u1));   // (now: hipcc otherwise keeps the raw vectors for it until after the products)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!KFULL) {   // coordinates beyond the rank read a neighbour's value: times 0
#pragma unroll
          for (int s2 = 0; s2 < 16; s2++) {
            xs0[s2] *= m0;
            xs1[s2] *= m1;
          }
        }
        {
          f16x8 h0, l0, h1, l1;
          mf_operands(xs0, h0, l0, h1, l1);
          stash(0, h0); stash(4, l0); stash(1, h1); stash(5, l1);
          __builtin_amdgcn_sched_barrier(0);
          mf_operands(xs1, h0, l0, h1, l1);
          stash(2, h0); stash(6, l0); stash(3, h1); stash(7, l1);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (!SYM) {
            // some confidence below 1 (sqrt(c - 1) undefined): the A side carries 2^e' (c - 1) 2^e x, the B side 2^e x
#pragma unroll
            for (int s2 = 0; s2 < 16; s2++) {
              const float w = readlane_f(wj, s2);
              xs0[s2] *= w;
              xs1[s2] *= w;
            }
            mf_operands(xs0, h0, l0, h1, l1);
            stash(8, h0); stash(12, l0); stash(9, h1); stash(13, l1);
            mf_operands(xs1, h0, l0, h1, l1);
            stash(10, h0); stash(14, l0); stash(11, h1); stash(15, l1);
          }
        }
        asm volatile("" : "+v"(idn), "+v"(cvn));   // (hipcc's wait for the two lands here, not behind their request)
      }
      products();
    }
    ln = lane;
    asm volatile("" : "+v"(ln));
    const int n = ln & 31, hf = ln >> 5;
    const bool lk0 = ln < k, lk1 = ln + 64 < k;
    // unscale, + XtX (implicit) / lambda_use I (explicit).  Tile (I, K), lane (n, hf), register v = [row 32 I + n]
    // [column 32 K + rho(v, hf)].  XtX comes padded to 128 x 128 (a.mf_XtX: identity on the padded diagonal; the operands of
    // the padded coordinates were zero), per lane = row with the columns at uniform offsets: no clamps, no selects (a
    // selected load is sunk under its condition, one round trip per entry)
    const float unscale = (un1 * un1) * unw;   // (powers of two: exact)
    MF_TICK(1)
    MF_DRAIN();
    mf_sfor<4>([&](auto it2) {
      constexpr int I = decltype(it2)::value;
      mf_sfor<I + 1>([&](auto kt) {
        constexpr int K = decltype(kt)::value;
        constexpr int T = mf_tid(I, K);
        float gv[16];
        if constexpr (IMPLICIT) {
          const float* gp = a.mf_XtX + (32 * I + n) + (4 * hf) * 128;
#pragma unroll
          for (int v = 0; v < 16; v++) gv[v] = gp[(32 * K + 8 * (v >> 2) + (v & 3)) * 128];
        } else {
          const float dg = (32 * I + n) < k ? lam_use : 1.f;
#pragma unroll
          for (int v = 0; v < 16; v++) gv[v] = (I == K && n == 8 * (v >> 2) + 4 * hf + (v & 3)) ? dg : 0.f;
        }
        mf_sfor<16>([&](auto vt) {
          constexpr int v = decltype(vt)::value;
          mf_wr<16 * T + v>(fmaf(mf_rd<16 * T + v>(), unscale, gv[v]));
        });
      });
    });
    // right-hand side per row: B[0][r] = rhs[r], B[1][r] = 0 (the halves carry partial sums from here on)
    wave_sync();
    sm.B[0][ln] = lk0 ? u0 : 0.f;
    sm.B[0][64 + ln] = lk1 ? u1 : 0.f;
    sm.B[1][ln] = 0.f;
    sm.B[1][64 + ln] = 0.f;
    wave_sync();

    // ---------------- blocked Cholesky, 16 panels of 8 columns ----------------
    MF_TICK(2)
    bool bad = false;
#ifdef RSP_MF_ABL   // timing-only dev builds: the assembly alone (bit 0: no factorisation, bit 1: no backward pass, bit 2: no loss pass)
    for (int p = 0; p < ((RSP_MF_ABL & 1) ? 0 : 16); p++) {
#else
    for (int p = 0; p < 16; p++) {
#endif
      const int K = p >> 2, q = p & 3;
      const int nsl = 4 - K;          // tiles (K + s, K), s < nsl
      const int thr = 8 * (q + 1);    // rows of tile (K, K) below the panel's diagonal block: n >= thr
      float P[4][4], Bs[4];
#pragma unroll
      for (int s = 0; s < 4; s++) {
#pragma unroll
        for (int m = 0; m < 4; m++) P[s][m] = 0.f;
      }
      MF_DRAIN();   // the previous panel's trailing updates have landed
      // (the tile of the diagonal block now, the tiles below it after the block is factored: 12 registers less while the
      // 36 of the block are live -- as one copy-out hipcc kept a slot in scratch)
#define MF_CALL(KK, QQ) mf_copy_out<KK, QQ, 0, 1>(P)
      MF_SWITCH16(p, MF_CALL)
#undef MF_CALL
      Bs[0] = sm.B[hf][32 * K + n];
      // the diagonal block and the right-hand side of its rows -> every lane
      wave_sync();
      if ((n >> 3) == q) {
        *reinterpret_cast<float4*>(&sm.D[(n & 7) * 8 + 4 * hf]) = make_float4(P[0][0], P[0][1], P[0][2], P[0][3]);
        sm.Bp[hf * 8 + (n & 7)] = Bs[0];
      }
      wave_sync();
      float L[8][8], inv[8], u8[8];
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const float4 lo = *reinterpret_cast<const float4*>(&sm.D[r * 8]);
        L[r][0] = lo.x; L[r][1] = lo.y; L[r][2] = lo.z; L[r][3] = lo.w;
        if (r >= 4) {
          const float4 hi = *reinterpret_cast<const float4*>(&sm.D[r * 8 + 4]);
          L[r][4] = hi.x; L[r][5] = hi.y; L[r][6] = hi.z; L[r][7] = hi.w;
        }
      }
      {
        const float4 b0 = *reinterpret_cast<const float4*>(&sm.Bp[0]), b1 = *reinterpret_cast<const float4*>(&sm.Bp[4]);
        const float4 c0 = *reinterpret_cast<const float4*>(&sm.Bp[8]), c1 = *reinterpret_cast<const float4*>(&sm.Bp[12]);
        u8[0] = b0.x + c0.x; u8[1] = b0.y + c0.y; u8[2] = b0.z + c0.z; u8[3] = b0.w + c0.w;
        u8[4] = b1.x + c1.x; u8[5] = b1.y + c1.y; u8[6] = b1.z + c1.z; u8[7] = b1.w + c1.w;
      }
      // 8 x 8 Cholesky, the same in every lane; the forward substitution of the block's right-hand side rides along
#pragma unroll
      for (int c = 0; c < 8; c++) {
        float d = L[c][c];
#pragma unroll
        for (int m = 0; m < c; m++) d = fmaf(-L[c][m], L[c][m], d);
        bad = bad || !(d > 0.f);
        const float r0 = __builtin_amdgcn_rsqf(d);
        const float h = 0.5f * d * r0;
        const float ri = fmaf(r0, fmaf(-h, r0, 0.5f), r0);   // one Newton step on 1 / sqrt(d)
        inv[c] = ri;
#pragma unroll
        for (int r = c + 1; r < 8; r++) {
          float v = L[r][c];
#pragma unroll
          for (int m = 0; m < c; m++) v = fmaf(-L[r][m], L[c][m], v);
          L[r][c] = v * ri;
        }
        float uv = u8[c];
#pragma unroll
        for (int m = 0; m < c; m++) uv = fmaf(-L[c][m], u8[m], uv);
        u8[c] = uv * ri;
      }
      if (ln == 0) {   // for the backward pass (diagonal entries: only their reciprocals are ever used)
        *reinterpret_cast<float4*>(&sm.U[8 * p]) = make_float4(u8[0], u8[1], u8[2], u8[3]);
        *reinterpret_cast<float4*>(&sm.U[8 * p + 4]) = make_float4(u8[4], u8[5], u8[6], u8[7]);
        *reinterpret_cast<float4*>(&sm.Inv[8 * p]) = make_float4(inv[0], inv[1], inv[2], inv[3]);
        *reinterpret_cast<float4*>(&sm.Inv[8 * p + 4]) = make_float4(inv[4], inv[5], inv[6], inv[7]);
#pragma unroll
        for (int r = 1; r < 8; r++) {
          *reinterpret_cast<float4*>(&sm.L8[p][r * 8]) =
              make_float4(L[r][0], r >= 2 ? L[r][1] : 0.f, r >= 3 ? L[r][2] : 0.f, r >= 4 ? L[r][3] : 0.f);
          if (r >= 5)
            *reinterpret_cast<float4*>(&sm.L8[p][r * 8 + 4]) =
                make_float4(L[r][4], r >= 6 ? L[r][5] : 0.f, r >= 7 ? L[r][6] : 0.f, 0.f);
        }
      }
      if (nsl > 1) {   // wave-uniform
#define MF_CALL(KK, QQ) mf_copy_out<KK, QQ, 1, 4>(P)
        MF_SWITCH16(p, MF_CALL)
#undef MF_CALL
      }
#pragma unroll
      for (int s = 1; s < 4; s++) Bs[s] = s < nsl ? sm.B[hf][32 * (K + s) + n] : 0.f;
      // every lane's row against the block: columns 0..3 at the lanes (n, 0), then -- with those -- columns 4..7 at (n, 1)
      const float uu[4] = {hf ? u8[4] : u8[0], hf ? u8[5] : u8[1], hf ? u8[6] : u8[2], hf ? u8[7] : u8[3]};
#pragma unroll
      for (int s = 0; s < 4; s++) {
        if (s < nsl) {   // wave-uniform
          const float x0 = P[s][0] * inv[0];
          const float x1 = fmaf(-x0, L[1][0], P[s][1]) * inv[1];
          const float x2 = fmaf(-x1, L[2][1], fmaf(-x0, L[2][0], P[s][2])) * inv[2];
          const float x3 = fmaf(-x2, L[3][2], fmaf(-x1, L[3][1], fmaf(-x0, L[3][0], P[s][3]))) * inv[3];
          const float xx[4] = {x0, x1, x2, x3};
          float t[4];
#pragma unroll
          for (int m = 0; m < 4; m++) {
            const unsigned xu = __float_as_uint(xx[m]);
            t[m] = __uint_as_float(__builtin_amdgcn_permlane32_swap(xu, xu, false, false)[0]);   // the value of the lane (n, 0)
          }
          float y[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            float v = P[s][j];
#pragma unroll
            for (int m = 0; m < 4; m++) v = fmaf(-t[m], L[4 + j][m], v);
#pragma unroll
            for (int m = 0; m < j; m++) v = fmaf(-y[m], L[4 + j][4 + m], v);
            y[j] = v * inv[4 + j];
          }
          const bool keep = s > 0 || n >= thr;
#pragma unroll
          for (int m = 0; m < 4; m++) {
            P[s][m] = keep ? (hf ? y[m] : xx[m]) : 0.f;
            Bs[s] = fmaf(-P[s][m], uu[m], Bs[s]);
          }
          sm.B[hf][32 * (K + s) + n] = Bs[s];
        }
      }
#define MF_CALL(KK, QQ) mf_copy_in_update<KK, QQ>(P)
      MF_SWITCH16(p, MF_CALL)
#undef MF_CALL
    }

    // ---------------- backward: L^T y = u, last panel first; y replaces u in sm.U ----------------
    MF_TICK(3)
    MF_DRAIN();
    wave_sync();
#ifdef RSP_MF_ABL
    for (int p = ((RSP_MF_ABL & 2) ? -1 : 15); p >= 0; p--) {
#else
    for (int p = 15; p >= 0; p--) {
#endif
      const int K = p >> 2, q = p & 3;
      const int nsl = 4 - K;
      const int thr = 8 * (q + 1);
      float P[4][4];
#pragma unroll
      for (int s = 0; s < 4; s++)
#pragma unroll
        for (int m = 0; m < 4; m++) P[s][m] = 0.f;
#define MF_CALL(KK, QQ) mf_copy_out<KK, QQ, 0, 4>(P)
      MF_SWITCH16(p, MF_CALL)
#undef MF_CALL
      float w[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; s++) {
        if (s < nsl) {
          const bool keep = s > 0 || n >= thr;
          const float yv = keep ? sm.U[32 * (K + s) + n] : 0.f;   // (rows at or above the block: not part of this sum)
#pragma unroll
          for (int m = 0; m < 4; m++) w[m] = fmaf(keep ? P[s][m] : 0.f, yv, w[m]);
        }
      }
      float S[8];
#pragma unroll
      for (int m = 0; m < 4; m++) {
        float v = w[m];
        v += dpp<0xB1>(v);   // quad_perm:[1,0,3,2]
        v += dpp<0x4E>(v);   // quad_perm:[2,3,0,1]
        v += dpp<0x141>(v);  // row_half_mirror
        v += dpp<0x140>(v);  // row_mirror: every lane of a row of 16 holds the row's sum
        S[m] = readlane_f(v, 0) + readlane_f(v, 16);
        S[4 + m] = readlane_f(v, 32) + readlane_f(v, 48);
      }
      float L[8][8], inv[8], y8[8];
#pragma unroll
      for (int r = 1; r < 8; r++) {
        const float4 lo = *reinterpret_cast<const float4*>(&sm.L8[p][r * 8]);
        L[r][0] = lo.x; L[r][1] = lo.y; L[r][2] = lo.z; L[r][3] = lo.w;
        if (r >= 5) {
          const float4 hi = *reinterpret_cast<const float4*>(&sm.L8[p][r * 8 + 4]);
          L[r][4] = hi.x; L[r][5] = hi.y; L[r][6] = hi.z; L[r][7] = hi.w;
        }
      }
      {
        const float4 i0 = *reinterpret_cast<const float4*>(&sm.Inv[8 * p]), i1 = *reinterpret_cast<const float4*>(&sm.Inv[8 * p + 4]);
        inv[0] = i0.x; inv[1] = i0.y; inv[2] = i0.z; inv[3] = i0.w; inv[4] = i1.x; inv[5] = i1.y; inv[6] = i1.z; inv[7] = i1.w;
        const float4 a0 = *reinterpret_cast<const float4*>(&sm.U[8 * p]), a1 = *reinterpret_cast<const float4*>(&sm.U[8 * p + 4]);
        y8[0] = a0.x; y8[1] = a0.y; y8[2] = a0.z; y8[3] = a0.w; y8[4] = a1.x; y8[5] = a1.y; y8[6] = a1.z; y8[7] = a1.w;
      }
#pragma unroll
      for (int c = 7; c >= 0; c--) {
        float v = y8[c] - S[c];
#pragma unroll
        for (int r = c + 1; r < 8; r++) v = fmaf(-L[r][c], y8[r], v);
        y8[c] = v * inv[c];
      }
      wave_sync();
      if (ln == 0) {
        *reinterpret_cast<float4*>(&sm.U[8 * p]) = make_float4(y8[0], y8[1], y8[2], y8[3]);
        *reinterpret_cast<float4*>(&sm.U[8 * p + 4]) = make_float4(y8[4], y8[5], y8[6], y8[7]);
      }
      wave_sync();
    }

    MF_TICK(4)
    if (bad) {   // wave-uniform: the general solver re-solves the row and owns its loss term (wrmf_lu.hip)
      int pos = 0;
      if (lane == 0) pos = atomicAdd(a.fail_counter, 1);
      pos = rfl(pos);
      if (pos < a.fail_cap) {
        if (lane == 0) a.fail_rows[pos] = row;
      } else {
        if (lk0) yrow[ln] = 0.f;   // no room in the list: unresolved, zeroed like a singular row
        if (lk1) yrow[64 + ln] = 0.f;
      }
      continue;
    }
    const float z0 = sm.U[ln], z1 = sm.U[64 + ln];
    if (lk0) yrow[ln] = z0;
    if (lk1) yrow[64 + ln] = z1;

    // ---------------- loss row term: lane j takes non-zero j of a chunk, y from LDS ----------------
    float lacc = 0.f;
#ifdef RSP_MF_ABL
    for (int base = p1; base < ((RSP_MF_ABL & 4) ? p1 : p2); base += 64) {
#else
    for (int base = p1; base < p2; base += 64) {
#endif
      const int ccnt = min(64, p2 - base);
      const int jl = min(lane, ccnt - 1);
      const float* xr = a.X + (size_t)a.row_idx[base + jl] * k;
      const float cvv = a.vals[base + jl];
      float t0 = 0.f, t1 = 0.f;
      if constexpr (KFULL) {
        // rank 128: the lane's vector in four batches of eight 16-byte loads, every load of a batch requested before the
        // first product (as one loop hipcc waited for each load: 50 k cycles per row for this pass)
#pragma unroll
        for (int hb = 0; hb < 4; hb++) {
          float4 xv[8];
#pragma unroll
          for (int q = 0; q < 8; q++) xv[q] = *reinterpret_cast<const float4*>(xr + 32 * hb + 4 * q);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 8; q++) {
            const float4 b = *reinterpret_cast<const float4*>(&sm.U[32 * hb + 4 * q]);
            t0 = fmaf(xv[q].x, b.x, t0);
            t1 = fmaf(xv[q].y, b.y, t1);
            t0 = fmaf(xv[q].z, b.z, t0);
            t1 = fmaf(xv[q].w, b.w, t1);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else if (vec) {
        for (int m4 = 0; m4 < k / 4; m4++) {
          const float4 xv = *reinterpret_cast<const float4*>(xr + 4 * m4);
          const float4 b = *reinterpret_cast<const float4*>(&sm.U[4 * m4]);
          t0 = fmaf(xv.x, b.x, t0);
          t1 = fmaf(xv.y, b.y, t1);
          t0 = fmaf(xv.z, b.z, t0);
          t1 = fmaf(xv.w, b.w, t1);
        }
      } else {
        for (int m = 0; m < k; m++) t0 = fmaf(xr[m], sm.U[m], t0);
      }
      const float tt = t0 + t1;
      const float d = IMPLICIT ? a.loss_tgt_const - tt : cvv - tt;
      lacc += lane < ccnt ? (IMPLICIT ? cvv * d * d : d * d) : 0.f;
    }
    const float lpart = wave_sum(lacc);
    const float xxp = wave_sum(z0 * z0 + z1 * z1);
    MF_TICK(5)
    if (lane == 0) sm.loss += IMPLICIT ? (double)lpart + a.lambda_loss * (double)xxp : (double)(lpart + lam_use * xxp);
    wave_sync();
  }
  if (lane == 0) a.loss_partials[loss_slot0 + blockIdx.x] = sm.loss;
#ifdef RSP_MF_PROF
  if (lane == 0 && a.ne_prof)
    for (int i = 0; i < 8; i++) atomicAdd(a.ne_prof + 8 + i, pt[i]);
#endif
}

}  // namespace
}  // namespace rsparse_hip

// the kernels by fixed names (wrmf_chol_mf.attrs.csv): implicit feedback with every confidence >= 1 (one operand set), implicit
// feedback in general, explicit feedback; each at rank 128 and at the ranks 65..127 (padded coordinates masked)
#define MF_KERNEL(NAME, IMP, SYM, KF)                                                                                      \
  extern "C" __global__ __launch_bounds__(64, MF_WAVES) __attribute__((amdgpu_num_vgpr(MF_NUM_VGPR))) void NAME(            \
      rsparse_hip::AlsArgs a, const int32_t* __restrict__ rows, int n_rows, int loss_slot0) {                               \
    rsparse_hip::als_chol_mf_body<IMP, SYM, KF>(a, rows, n_rows, loss_slot0);                                               \
  }
MF_KERNEL(rsparse_hip_als_chol_mf_implicit, true, true, true)
MF_KERNEL(rsparse_hip_als_chol_mf_implicit_any, true, false, true)
MF_KERNEL(rsparse_hip_als_chol_mf_explicit, false, true, true)
MF_KERNEL(rsparse_hip_als_chol_mf_implicit_padded, true, true, false)
MF_KERNEL(rsparse_hip_als_chol_mf_implicit_any_padded, true, false, false)
MF_KERNEL(rsparse_hip_als_chol_mf_explicit_padded, false, true, false)
#undef MF_KERNEL

namespace rsparse_hip {

bool chol_mf_supported(int k) { return padded_rank(k) == 128 && k > 64; }
int chol_mf_grid(int n_rows) { return std::max(1, std::min(n_rows, kCholMfGrid)); }
// implicit feedback: two launches (see the kernel's first lines), each with its own slots
int chol_mf_loss_slots(int n_rows, bool implicit) { return (implicit ? 2 : 1) * chol_mf_grid(n_rows); }

// rows[0 .. n_rows): the rows of 65 .. kCholMfMax non-zeros; loss partials [loss_slot0, loss_slot0 + chol_mf_grid(n_rows))
hipError_t launch_als_chol_mf(const AlsArgs& a, bool implicit, const int32_t* rows, int n_rows, int loss_slot0, hipStream_t s,
                              hipEvent_t* ev_slot) {
  if (n_rows <= 0) return hipSuccess;
  if (!a.wave_stats || !chol_mf_supported(a.k)) return hipErrorInvalidValue;
  const int grid = chol_mf_grid(n_rows);
#ifdef RSP_MF_PROF
  {
    int nb = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rsparse_hip_als_chol_mf_implicit, 64, 0);
    hipFuncAttributes fa{};
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(rsparse_hip_als_chol_mf_implicit));
    std::fprintf(stderr, "[mf_prof] occupancy %d workgroups of one wave per CU; %d registers, %zu bytes of LDS, %zu of scratch\n", nb,
                 fa.numRegs, fa.sharedSizeBytes, fa.localSizeBytes);
  }
#endif
  const bool full = a.k == 128 && (reinterpret_cast<uintptr_t>(a.X) & 15) == 0;   // (the full kernels read 16 bytes at a time in their loss pass)
  if (implicit) {
    // (both: the device-side flag decides; the symmetric one is the normal case and is named for the profile)
    auto kern = full ? rsparse_hip_als_chol_mf_implicit : rsparse_hip_als_chol_mf_implicit_padded;
    auto kern2 = full ? rsparse_hip_als_chol_mf_implicit_any : rsparse_hip_als_chol_mf_implicit_any_padded;
    prof_note(ev_slot, reinterpret_cast<const void*>(kern));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, s, a, rows, n_rows, loss_slot0);
    hipLaunchKernelGGL(kern2, dim3(grid), dim3(64), 0, s, a, rows, n_rows, loss_slot0 + grid);
  } else {
    auto kern = full ? rsparse_hip_als_chol_mf_explicit : rsparse_hip_als_chol_mf_explicit_padded;
    prof_note(ev_slot, reinterpret_cast<const void*>(kern));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, s, a, rows, n_rows, loss_slot0);
  }
  return hipGetLastError();
}

}  // namespace rsparse_hip
