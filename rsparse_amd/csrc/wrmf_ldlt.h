// k x k LDL^T solve by one workgroup of four waves, the matrix in REGISTERS (gfx950, wave64).
//
// The exact solve of the normal-equation kernel (wrmf_ne.hip, solver == CHOLESKY: y = solve(lhs, rhs),
// wrmf_implicit.hpp:231,236 / wrmf_explicit.hpp:103-108).  Layout: lane = row (rows lane and lane + 64 at rank 128),
// register = column; the 8-wide block columns are dealt round robin to the waves (wave w owns blocks w, w + 4, ...), so a
// wave holds 48 registers of the lower triangle and every rank-1 step is a lane-parallel FMA whose multiplier is a lane
// broadcast: no 16-lane groups computing the same thing, no transposed reads, no tile traffic through LDS.
//   forward, block column J (owner = wave J % 4):
//     factor: right-looking inside the block, the pivot row's entries by v_readlane; the forward substitution of the
//       right-hand side rides along (it lives in LDS between owners); the block's columns are published twice:
//       l_ij by columns (the vector operand of the others' updates) and d_j l_cj by rows (their broadcast operand);
//     update: every wave subtracts the panel from the blocks it still owns; the owner of block J + 1 does that block
//       first and factors it at once (look-ahead), the others' updates run beside it.  One barrier per block column.
//   backward, last block first: the owner dots its columns with the finished part of y (a transposed wave reduction),
//     then solves the 8 x 8 triangle with multipliers from the copy of the diagonal blocks kept in LDS.
// What bounds it (tools/probes/valu_rate_probe.hip): one wave issues a vector instruction every ~4.3 cycles (4.8 with a DPP
// operand, 8.4 for v_readlane), whatever the other waves of the SIMD do -- so the solve time is the instruction count of
// its critical path (update of the next block + its factorisation, once per block column), and narrow blocks halve that
// against 16-wide ones; uniform operands read from LDS would move 64 copies each through the LDS return path.
// A pivot that is not positive makes solve() return true (the caller hands the row to wrmf_lu.hip).
#pragma once
#include <type_traits>
#include <utility>

#include "wrmf_device.h"

#ifdef LDLT_PROF   // tools/probes/ldlt_probe.hip: ticks per phase and wave
#define LDLT_T(j) { const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); ldlt_prof[j] += t1_ - ldlt_tl; ldlt_tl = t1_; }
#else
#define LDLT_T(j)
#endif

namespace rsparse_hip {
namespace dev {

template <int KP, int TLD>
struct Ldlt {
  static constexpr int BC = 8;             // columns per block
  static constexpr int RH = KP / 64;       // rows per lane
  static constexpr int NBLK = KP / BC;     // block columns
  static constexpr int NOWN = NBLK / 4;    // blocks per wave (2: rank 64, 4: rank 128)
  static constexpr int BPH = 64 / BC;      // blocks per 64 rows
  static constexpr int TS = 32 * TLD;
  static constexpr int NBUF = KP == 128 ? 3 : 2;   // panels in flight (rank 64: the scratch must fit the three tiles it overlays)
  static constexpr int PS_FLOATS = NBUF * BC * KP;   // [NBUF][BC][KP]  l_ij, column-major (lane-contiguous)
  static constexpr int PU_FLOATS = PS_FLOATS;        // [NBUF][BC][KP]  d_j l_ij, the same layout
  static constexpr int DG_FLOATS = NBLK * BC * BC;   // [NBLK][BC][BC] diagonal blocks of L, [row][column]
  static constexpr int FLOATS = PS_FLOATS + PU_FLOATS + DG_FLOATS;
  static_assert(KP == 64 || KP == 128, "rank padded to 64 or 128");
  // slot s of wave w = block blk(w, s); the 64-row half that holds its diagonal block, and the halves it has registers for
  // (plain round robin.  Dealing the groups of four alternately forwards and backwards gives every wave the same number of
  //  updates, but then one wave owns blocks 3 and 4, 7 and 8, ...: its catching up sits on the critical path; measured 9 % slower)
  __host__ __device__ static constexpr int owner(int J) { return J & 3; }
  __host__ __device__ static constexpr int blk(int w, int s) { return 4 * s + w; }
  __host__ __device__ static constexpr int hd(int s) { return (4 * s) / BPH; }
  __host__ __device__ static constexpr int nh(int s) { return RH - hd(s); }

  template <class F, int... I>
  static __device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
  }
  template <int N, class F>
  static __device__ __forceinline__ void sfor(F&& f) {
    sfor_impl(f, std::make_integer_sequence<int, N>{});
  }

  template <int E>
  static __device__ __forceinline__ void fnma_bcast(float& acc, const float u, const float l) { fnma_row_bcast<E>(acc, u, l); }
  static __device__ __forceinline__ float swap32_add(float a, float b) {   // lanes < 32: sum of a's halves, >= 32: of b's
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  static __device__ __forceinline__ float swap16_add(float a, float b) {   // rows 0, 2: a's rows 0+1, 2+3; rows 1, 3: b's
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  static __device__ __forceinline__ float row_sum(float v) {   // sum over each row of 16 lanes, in all of its lanes
    v += dpp<0xB1>(v);
    v += dpp<0x4E>(v);
    v += dpp<0x141>(v);
    v += dpp<0x140>(v);
    return v;
  }

  // Flags in LDS between the waves.  The LDS executes one wave's instructions in order, so a flag written after the data
  // is seen after the data, and data read after the flag is the data the flag announced: no s_waitcnt on either side,
  // only compiler fences.
  static __device__ __forceinline__ void lds_post(int* p, const int v) {
    asm volatile("" ::: "memory");
    *reinterpret_cast<volatile int*>(p) = v;
    asm volatile("" ::: "memory");
  }
  static __device__ __forceinline__ void lds_wait_ge(const int* p, const int v) {
    while (*reinterpret_cast<const volatile int*>(p) < v) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
  }

  // Block column J (first row in lane jl0 of half hd) of the wave's registers `col`; NH halves are held, index 0 is the
  // half with the diagonal block.  The pivots are a serial chain (broadcast the pivot, reciprocal, scale the column, update
  // the next column, broadcast ...); a wave issues in order, so the lane broadcasts of a column are all read as soon as
  // the column is final and the remaining rank-1 updates of the column before it sit between the links of the chain.
  template <int NH>
  static __device__ __forceinline__ void factor(float (&col)[BC][RH], const int J, const int jl0, const int pbuf, float* sCh,
                                                float* sU, int* sRdy, bool& bad, const int lane) {
    constexpr int hoff = RH - NH;
    float u[NH];
#pragma unroll
    for (int x = 0; x < NH; x++) u[x] = sU[lane + 64 * (x + hoff)];
    float* ps = sCh + pbuf * BC * KP + lane + 64 * hoff;
    float* pu = ps + PS_FLOATS;
    float inv, l[NH], sc[BC];
    {
      const float pj = readlane_f(col[0][0], jl0);
#pragma unroll
      for (int c2 = 1; c2 < BC; c2++) sc[c2] = readlane_f(col[0][0], jl0 + c2);   // d_j l_{c2 j} of column j = 0
      if (!(pj > 0.f)) bad = true;
      inv = __builtin_amdgcn_rcpf(pj);   // (1 ulp; the quotients l_ij are then good to 1.5 ulp)
      l[0] = lane > jl0 ? col[0][0] * inv : 0.f;
      if constexpr (NH == 2) l[1] = col[0][1] * inv;
    }
    sfor<BC>([&](auto cct) {
      constexpr int cc = decltype(cct)::value;
      const int jl = jl0 + cc;
      auto upd = [&](const int c2) {   // rank-1 update of column c2 by column cc
#pragma unroll
        for (int x = 0; x < NH; x++) col[c2][x] = fmaf(-l[x], sc[c2], col[c2][x]);
      };
      constexpr int NF = BC - 2 - cc > 0 ? BC - 2 - cc : 0;   // updates of columns cc + 2 .. BC - 1
      constexpr int FA = NF / 2;
      float invn = 0.f, ln[NH], sn[BC];
#pragma unroll
      for (int x = 0; x < NH; x++) ln[x] = 0.f;
      const float uj = readlane_f(u[0], jl);
#pragma unroll
      for (int x = 0; x < NH; x++) {
        pu[cc * KP + 64 * x] = col[cc][x];
        ps[cc * KP + 64 * x] = l[x];
      }
      if constexpr (cc + 1 < BC) {
        upd(cc + 1);
        const float pjn = readlane_f(col[cc + 1][0], jl + 1);
#pragma unroll
        for (int c2 = cc + 2; c2 < BC; c2++) sn[c2] = readlane_f(col[cc + 1][0], jl0 + c2);
#pragma unroll
        for (int f = 0; f < FA; f++) upd(cc + 2 + f);
        if (!(pjn > 0.f)) bad = true;
        invn = __builtin_amdgcn_rcpf(pjn);
#pragma unroll
        for (int f = FA; f < NF; f++) upd(cc + 2 + f);
      }
      u[0] = lane == jl ? uj * inv : fmaf(-l[0], uj, u[0]);   // row j keeps z_j / d_j for the backward pass
      if constexpr (NH == 2) u[1] = fmaf(-l[1], uj, u[1]);
      if constexpr (cc + 1 < BC) {
        ln[0] = lane > jl + 1 ? col[cc + 1][0] * invn : 0.f;
        if constexpr (NH == 2) ln[1] = col[cc + 1][1] * invn;
      }
#pragma unroll
      for (int x = 0; x < NH; x++) {
        col[cc][x] = l[x];
        l[x] = ln[x];
      }
#pragma unroll
      for (int c2 = cc + 2; c2 < BC; c2++) sc[c2] = sn[c2];
      inv = invn;
      if constexpr (cc == BC / 2 - 1) lds_post(sRdy, 2 * J + 1);   // the first half of the panel is out
    });
#pragma unroll
    for (int x = 0; x < NH; x++) sU[lane + 64 * (x + hoff)] = u[x];
    lds_post(sRdy, 2 * J + 2);
    if ((unsigned)(lane - jl0) < (unsigned)BC) {   // the diagonal block, [row][column], for the backward pass
      float* dg = sCh + 2 * PS_FLOATS + J * BC * BC + (lane - jl0) * BC;
#pragma unroll
      for (int cc = 0; cc < BC; cc++) dg[cc] = col[cc][0];
    }
  }

  // Block `Jown` of the wave (NH halves held) -= columns [4 H, 4 H + 4) of the panel in buffer pbuf:
  //   col[c2][x] -= l_{row, e} * (d_e l_{c2 e}),   the first factor a register per panel column (this lane's rows), the second
  // ONE register per panel column too (lane c2 of every 16-lane row = the value for block row c2), broadcast inside the FMA
  // (row_newbcast).  Both are lane-contiguous reads of the column-major panel copies.
  template <int NH, int H>
  static __device__ __forceinline__ void update_half(float (&col)[BC][RH], const int Jown, const int pbuf, const float* sCh,
                                                     const int lane) {
    constexpr int hoff = RH - NH;
    const float* ps = sCh + pbuf * BC * KP + 4 * H * KP;
    const float* pl = ps + lane + 64 * hoff;
    const float* pq = ps + PS_FLOATS + BC * Jown + (lane & 7);
    float ue[4], lv[4][NH];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      ue[e] = pq[e * KP];
#pragma unroll
      for (int x = 0; x < NH; x++) lv[e][x] = pl[e * KP + 64 * x];
    }
    // (consecutive FMAs go to different accumulators: a dependent one would wait for its predecessor's result)
#pragma unroll
    for (int e = 0; e < 4; e++)
      sfor<BC>([&](auto ct) {
        constexpr int c2 = decltype(ct)::value;
#pragma unroll
        for (int x = 0; x < NH; x++) fnma_bcast<c2>(col[c2][x], ue[e], lv[e][x]);
      });
  }

  template <int NH>
  static __device__ __forceinline__ void back(const float (&col)[BC][RH], const int J, const float* sCh, float* sU,
                                              const int lane) {
    constexpr int hoff = RH - NH;
    // After the transposed reduction the sum of column c = 4 m + (g & 1) * 2 + (g >> 1) sits in register m, row g of 16
    // lanes; lane (g, q) takes m = (q >> 2) & 1: column cl(lane), every column in four lanes of one row
    const int g = lane >> 4;
    const int cl = 4 * ((lane >> 2) & 1) + ((g & 1) * 2 + (g >> 1));
    float t = sU[BC * J + cl];   // z_j / d_j of the block
    if (J + 1 < NBLK) {
      float yv[NH];
#pragma unroll
      for (int x = 0; x < NH; x++) {
        const int row = lane + 64 * (x + hoff);
        yv[x] = row >= BC * (J + 1) ? sU[row] : 0.f;
      }
      float p[BC];
#pragma unroll
      for (int cc = 0; cc < BC; cc++) {
        p[cc] = col[cc][0] * yv[0];
        if constexpr (NH == 2) p[cc] = fmaf(col[cc][1], yv[1], p[cc]);
      }
      float q4[4], r2[2];
#pragma unroll
      for (int i = 0; i < 4; i++) q4[i] = swap32_add(p[2 * i], p[2 * i + 1]);       // lanes < 32: column 2 i, else 2 i + 1
#pragma unroll
      for (int m = 0; m < 2; m++) r2[m] = row_sum(swap16_add(q4[2 * m], q4[2 * m + 1]));   // rows: columns 4m, 4m+2, 4m+1, 4m+3
      t -= (lane & 4) ? r2[1] : r2[0];
    }
    const float* dg = sCh + 2 * PS_FLOATS + J * BC * BC + cl;
    float ld[BC];
#pragma unroll
    for (int c2 = 1; c2 < BC; c2++) ld[c2] = dg[c2 * BC];   // l_{c2, cl}: zero for cl >= c2
#pragma unroll
    for (int c2 = BC - 1; c2 >= 1; c2--) {
      // column c2 lives in row g2 = (c2 & 1) * 2 + ((c2 >> 1) & 1), lanes with (q >> 2) & 1 == c2 >> 2
      const float yb = readlane_f(t, 16 * ((c2 & 1) * 2 + ((c2 >> 1) & 1)) + 4 * (c2 >> 2));
      t = fmaf(-ld[c2], yb, t);
    }
    if ((lane & 11) == 0) sU[BC * J + cl] = t;   // one lane per column (q = 0 or 4)
  }

  // sA: the system as lower-triangular 32 x 32 tiles (tile (R, C) at (R (R + 1) / 2 + C) * 32 * TLD, row stride TLD; diagonal
  // tiles complete), sU: right-hand side -> solution, sCh: FLOATS of scratch that MAY overlap sA (the tiles are dead after
  // the first barrier in here).  Called by all four waves; the caller has synchronised the tiles and sU.
  static __device__ __forceinline__ bool solve(const float* sA, float* sCh, float* sU, int* sFlag, const int wv,
                                               const int lane
#ifdef LDLT_PROF
                                               , unsigned long long* ldlt_prof
#endif
  ) {
#ifdef LDLT_PROF
    unsigned long long ldlt_tl = __builtin_amdgcn_s_memtime();
#endif
    float c[NOWN][BC][RH];
    sfor<NOWN>([&](auto st) {
      constexpr int s = decltype(st)::value;
      const int c0 = BC * blk(wv, s);   // first column of the block
      const int C = c0 >> 5, cin = c0 & 31;
#pragma unroll
      for (int x = 0; x < nh(s); x++) {
        const int R = max((lane >> 5) + 2 * (x + hd(s)), C);   // (rows above the block: any finite value)
        const float* p = sA + (R * (R + 1) / 2 + C) * TS + (lane & 31) * TLD + cin;
#pragma unroll
        for (int cc = 0; cc < BC; cc++) c[s][cc][x] = p[cc];
      }
    });
    int* sRdy = sFlag + 1;    // half panels published (2 J + 2 = panel J complete)
    int* sDone = sFlag + 2;   // [4] panels a wave has applied to all of its blocks
    if (wv == 0 && lane < 6) sFlag[lane] = 0;
    LDLT_T(0)
    __syncthreads();
    LDLT_T(1)
    // Forward pass as a data flow, no barriers: panel J goes to buffer J % NBUF and is announced in sRdy, half by half; a
    // wave applies the panels in order.  The owner of block J + 1 applies panel J to that block only -- its first half while
    // block J's owner is still busy with the second --, factors it, announces it, and catches up on its other blocks
    // afterwards: what the other waves wait for is never behind work nobody needs yet.
    bool bad = false;
    int deferred = -1;   // panel this wave has applied to its next block only
    auto apply = [&](const int P, const int above) {   // panel P -> this wave's blocks beyond block `above`
      sfor<NOWN>([&](auto st) {
        constexpr int s = decltype(st)::value;
        const int Js = blk(wv, s);
        if (Js > above) {
          update_half<nh(s), 0>(c[s], Js, P % NBUF, sCh, lane);
          update_half<nh(s), 1>(c[s], Js, P % NBUF, sCh, lane);
        }
      });
      lds_post(sDone + wv, P + 1);
    };
    for (int J = 0; J < NBLK; J++) {
      if (wv == owner(J)) {
        if (J >= NBUF) {   // the buffer is free once every wave is through with panel J - NBUF
#pragma unroll
          for (int w2 = 0; w2 < 4; w2++) lds_wait_ge(sDone + w2, J - NBUF + 1);
        }
        LDLT_T(6)
        __builtin_amdgcn_s_setprio(3);
        sfor<NOWN>([&](auto st) {
          constexpr int s = decltype(st)::value;
          if ((J >> 2) == s) factor<nh(s)>(c[s], J, (BC * J) & 63, J % NBUF, sCh, sU, sRdy, bad, lane);
        });
        __builtin_amdgcn_s_setprio(0);
        if (bad) *sFlag = 1;
        LDLT_T(3)
        if (deferred >= 0) {
          apply(deferred, J);
          deferred = -1;
        }
        LDLT_T(4)
      }
      if (J + 1 == NBLK) break;
      if (wv == owner(J + 1)) {
        __builtin_amdgcn_s_setprio(3);
        sfor<NOWN>([&](auto st) {
          constexpr int s = decltype(st)::value;
          if (((J + 1) >> 2) == s) {
            lds_wait_ge(sRdy, 2 * J + 1);
            LDLT_T(6)
            update_half<nh(s), 0>(c[s], J + 1, J % NBUF, sCh, lane);
            LDLT_T(2)
            lds_wait_ge(sRdy, 2 * J + 2);
            LDLT_T(6)
            update_half<nh(s), 1>(c[s], J + 1, J % NBUF, sCh, lane);
            LDLT_T(2)
          }
        });
        deferred = J;
      } else {
        lds_wait_ge(sRdy, 2 * J + 2);
        LDLT_T(6)
        apply(J, J);
        LDLT_T(4)
      }
    }
    __syncthreads();
    LDLT_T(6)
    for (int J = NBLK - 1; J >= 0; J--) {
      if (wv == owner(J)) {
        __builtin_amdgcn_s_setprio(3);
        sfor<NOWN>([&](auto st) {
          constexpr int s = decltype(st)::value;
          if ((J >> 2) == s) back<nh(s)>(c[s], J, sCh, sU, lane);
        });
        __builtin_amdgcn_s_setprio(0);
      }
      LDLT_T(7)
      __syncthreads();
      LDLT_T(8)
    }
    return *sFlag != 0;
  }
};

}  // namespace dev
}  // namespace rsparse_hip
