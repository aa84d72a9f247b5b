// Exact (Cholesky) half-iteration at rank <= 64: one WAVE per row, the row's k x k system in the wave's registers
// (gfx950, wave64; round 4).
//
// (Rank 33..64: the assembly itself runs on the matrix cores, 16 non-zeros per instruction, see "assembly" below.)
// Replaces, for these ranks, the main launch of wrmf_chol.hip -- the solver == CHOLESKY branch of als_implicit<T> /
// als_explicit<T> (inst/include/wrmf_implicit.hpp:207-208,231,236; wrmf_explicit.hpp:103-108):
//     lhs = XtX + X_nnz diag(c - 1) X_nnz^T   |   X_nnz X_nnz^T + lambda_use I,      rhs = X_nnz c (+ rhs_init),
//     Y_new = solve(lhs, rhs, fast + likely_sympd).
// The 256-thread kernel there spends its time waiting (SQ counters on config 5, rank 64: waves 21 % issuing, 69 % waiting:
// a dozen barriers per block column for a 64 x 64 system that one wave can hold).  Here lane l holds row l of lhs in KP
// registers from the assembly (every gathered vector is broadcast through 256 bytes of LDS: 16 reads and 64 FMAs per
// non-zero) through a right-looking LDL^T in which the pivot row reaches the lanes as a DPP row broadcast inside the FMA
// (the register solve of wrmf_chol_lr.hip, with a guarded pivot) to both substitutions: no barrier, no matrix in LDS, twelve
// rows in flight per CU.  A non-positive pivot sends the row to the general solver (wrmf_lu.hip), exactly as the k x k
// kernel does.  Rows: the ones the main launch of wrmf_chol.hip would take (everything the normal-equation launch and the
// two-level LONG launch do not own).
#include <utility>

#include "wrmf_internal.h"
#include "wrmf_device.h"

namespace rsparse_hip {
namespace {

using namespace dev;

template <class F, int... I>
__device__ __forceinline__ void cw_sfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void cw_sfor(F&& f) {
  cw_sfor_impl(f, std::make_integer_sequence<int, N>{});
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// x (already scaled into fp16's range) -> fl16(x), fl16(x - fl16(x)) for a pair; the residual is exact in fp32
__device__ __forceinline__ void cw_split(const float x0, const float x1, unsigned& hi, unsigned& lo) {
  const f32x2 v = {x0, x1};
  const f16x2 h = __builtin_convertvector(v, f16x2);
  const f32x2 r = v - __builtin_convertvector(h, f32x2);
  const f16x2 l = __builtin_convertvector(r, f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ f16x8 cw_pack(const unsigned a, const unsigned b, const unsigned c, const unsigned d) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = {a, b, c, d};
  return __builtin_bit_cast(f16x8, v);
}
// biased exponent e of the power of two that brings `vmax` into [2^13, 2^14); 2^(e - 127) is the scale
__device__ __forceinline__ int cw_scale_exp(float vmax) {
  const int eb = (int)((__float_as_uint(vmax) >> 23) & 0xffu);
  return min(253, max(1, 267 - eb));
}
__device__ __forceinline__ float cw_pow2(int biased) { return __uint_as_float((unsigned)biased << 23); }

template <int KP, bool IMPLICIT>
__global__ __launch_bounds__(64, 3) void als_chol_wave_kernel(AlsArgs a, int loss_slot0) {
  // (one wave per WORKGROUP: with four, a workgroup held its slot until the longest of its four rows was done)
  __shared__ __attribute__((aligned(16))) float sXv[KP];   // the vector being broadcast
  const int lane = threadIdx.x & 63;
  float* sX = sXv;
  const int k = a.k;
  const bool on = lane < KP, lk = lane < k;
  const bool vec = (k % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.X) & 15) == 0);
  double wloss = 0.0;

  // the rows of the main launch as ranges of the length-sorted order (see als_chol2_kernel)
  const bool listed = a.chol_list != nullptr;
  // (explicit feedback with the push-through kernel of wrmf_chol_lr.hip: the rows of 1..64 ratings are its, the tail is the empty rows)
  const int tail0 = listed ? (a.lrx ? a.chol_empty_first : a.chol_first + a.chol_n_main) : 0;
  const int n_tail = listed ? a.n_cols - tail0 : 0;
  const int n_iter = listed ? a.chol_n_main + n_tail : a.n_cols;
  auto row_at = [&](const int it2) {
    if (!listed) return it2;
    return (int)a.chol_list[it2 < a.chol_n_main ? a.chol_first + it2 : tail0 + (it2 - a.chol_n_main)];
  };

  for (int it = blockIdx.x; it < n_iter; it += gridDim.x) {
    const int row = rfl(row_at(it));
    const int p1 = rfl(a.col_ptrs[row]), p2 = rfl(a.col_ptrs[row + 1]);
    const int cnt = p2 - p1;
    if (a.n_chol_long > 0 && cnt > kCholLongLen) continue;   // the LONG launch owns it
    if (a.ne_chol && cnt > a.ne_chol_min) continue;          // assembled and solved by wrmf_ne.hip
    float* yrow = a.Y + (size_t)row * k;
    if (cnt <= 0 && !a.rhs_init) {  // empty column -> zeros (wrmf_implicit.hpp:281, wrmf_explicit.hpp:142)
      if (lk) yrow[lane] = 0.f;
      continue;
    }
    const float lam_use =
        IMPLICIT ? 0.f : (float)(a.lambda_loss * (a.dynamic_lambda ? (double)(float)cnt : 1.0));
    // the lane id as this row sees it (hipcc otherwise hoists the lane-dependent addresses and compares of the unrolled
    // loops out of the row loop and spills them)
    int ln = lane;
    asm volatile("" : "+v"(ln));

    // ---------------- assembly: lane l holds lhs(l, :) and rhs_l ----------------
    float r[KP];
    float u = 0.f;
    if constexpr (KP == 64) {
      {
        // On the matrix cores (round 4): X_nnz diag(w) X_nnz^T is a product over the NON-ZEROS -- the lane = coordinate layout in
        // which the vectors arrive (lane l reads element l of every vector: coalesced) is the operand layout already, 16
        // non-zeros per instruction, once the two halves of the wave have traded eight registers: lanes (m, 0) then hold the
        // coordinates 0..31 of the non-zeros 0..7 and lanes (m, 1) those of the non-zeros 8..15 (block 0; block 1 likewise).
        // Operands as two fp16 terms of 2^e x (and of 2^e' w x), three products of order < 2 (2^-21 per product, as in
        // wrmf_ne.hip); the accumulator tiles (lane: column, registers: rows) are, the matrix being symmetric, rows spread over
        // the lane pair (n, n + 32): one lane swap per register pair hands every lane its row (as in wrmf_chol_lr.hip).
        // 64 FMAs + 16 LDS reads per non-zero become ~ 15 instructions.
        const int ex = cw_scale_exp(fmaxf(__uint_as_float(a.wave_stats[0]), 1e-30f));
        const float wmax = IMPLICIT ? fmaxf(__uint_as_float(a.wave_stats[1]) - 1.f, 1.f) : 1.f;
        const int ewb = (int)((__float_as_uint(wmax) >> 23) & 0xffu);   // 2^(127 - ewb - 1 + 127...) below: |w| 2^-(ewb - 126) <= 1
        const float sx = cw_pow2(ex), sw = cw_pow2(min(253, max(1, 253 - ewb)));   // sw = 2^(126 - ewb) <= 1 / wmax
        f32x16 t[2][2];
#pragma unroll
        for (int e = 0; e < 16; e++) t[0][0][e] = t[0][1][e] = t[1][0][e] = t[1][1][e] = 0.f;
        const int lnc = min(ln, k - 1);
        auto fetch = [&](const int base, float (&xs)[16], int& idj, float& cvj, float& rcj, int& ccnt) {
          ccnt = min(16, p2 - base);
          if (ccnt <= 0) return;
          const int jl = min(lane & 15, ccnt - 1);
          idj = a.row_idx[base + jl];
          cvj = a.vals[base + jl];
          rcj = a.rhs_vals ? a.rhs_vals[base + jl] : cvj;
#pragma unroll
          for (int s2 = 0; s2 < 16; s2++) {
            const int id = __builtin_amdgcn_readlane(idj, min(s2, ccnt - 1));
            const float v = a.X[(size_t)id * k + lnc];   // (unconditional: a select, not a branch per load)
            xs[s2] = lk ? v : 0.f;
          }
        };
        float xs[16], xn[16];
        int idj = 0, idn = 0, ccnt = 0, cnn = 0;
        float cvj = 0.f, rcj = 0.f, cvn = 0.f, rcn = 0.f;
        fetch(p1, xs, idj, cvj, rcj, ccnt);
        for (int base = p1; base < p2; base += 16) {
          cnn = 0;
          if (base + 16 < p2) fetch(base + 16, xn, idn, cvn, rcn, cnn);
          // scaled operands; a slot beyond the row gets the scale 0
          float xa[16], xb[16];   // A side: w x (implicit) / x; B side: x
#pragma unroll
          for (int s2 = 0; s2 < 16; s2++) {
            const bool in = s2 < ccnt;   // wave-uniform
            const float rc = readlane_f(rcj, s2), cv = readlane_f(cvj, s2);
            if (in) u = fmaf(rc, xs[s2], u);
            const float sc = in ? sx : 0.f;
            xb[s2] = xs[s2] * sc;
            if constexpr (IMPLICIT) xa[s2] = xb[s2] * ((cv - 1.f) * sw);
          }
          f16x8 bh[2], bl[2], ah[2], al[2];
          {
            unsigned h0[4], l0[4], h1[4], l1[4];
            float b0[8], b1[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {   // block 0 <- (own, partner's) registers e / 8 + e; block 1 likewise
              const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(xb[e]), __float_as_uint(xb[8 + e]), false, false);
              b0[e] = __uint_as_float(sw2[0]);
              b1[e] = __uint_as_float(sw2[1]);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
              cw_split(b0[2 * q], b0[2 * q + 1], h0[q], l0[q]);
              cw_split(b1[2 * q], b1[2 * q + 1], h1[q], l1[q]);
            }
            bh[0] = cw_pack(h0[0], h0[1], h0[2], h0[3]); bl[0] = cw_pack(l0[0], l0[1], l0[2], l0[3]);
            bh[1] = cw_pack(h1[0], h1[1], h1[2], h1[3]); bl[1] = cw_pack(l1[0], l1[1], l1[2], l1[3]);
            if constexpr (IMPLICIT) {
#pragma unroll
              for (int e = 0; e < 8; e++) {
                const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(xa[e]), __float_as_uint(xa[8 + e]), false, false);
                b0[e] = __uint_as_float(sw2[0]);
                b1[e] = __uint_as_float(sw2[1]);
              }
#pragma unroll
              for (int q = 0; q < 4; q++) {
                cw_split(b0[2 * q], b0[2 * q + 1], h0[q], l0[q]);
                cw_split(b1[2 * q], b1[2 * q + 1], h1[q], l1[q]);
              }
              ah[0] = cw_pack(h0[0], h0[1], h0[2], h0[3]); al[0] = cw_pack(l0[0], l0[1], l0[2], l0[3]);
              ah[1] = cw_pack(h1[0], h1[1], h1[2], h1[3]); al[1] = cw_pack(l1[0], l1[1], l1[2], l1[3]);
            } else {
              ah[0] = bh[0]; al[0] = bl[0]; ah[1] = bh[1]; al[1] = bl[1];
            }
          }
#pragma unroll
          for (int ta = 0; ta < 2; ta++)
#pragma unroll
            for (int tb = 0; tb < 2; tb++) {
              t[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ta], bh[tb], t[ta][tb], 0, 0, 0);
              t[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ta], bl[tb], t[ta][tb], 0, 0, 0);
              t[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ta], bh[tb], t[ta][tb], 0, 0, 0);
            }
#pragma unroll
          for (int s2 = 0; s2 < 16; s2++) xs[s2] = xn[s2];
          idj = idn; cvj = cvn; rcj = rcn; ccnt = cnn;
        }
        // rows: tile (a, b) at lane (n, hf), register v = lhs[32 a + rho(v, hf)][32 b + n] = row 32 b + n, column 32 a + rho(v, hf)
        const float un1 = cw_pow2(254 - ex), un2 = IMPLICIT ? un1 * (wmax >= 1.f ? 1.f : 1.f) : un1;
        const float unw = IMPLICIT ? cw_pow2(254 - min(253, max(1, 253 - ewb))) : 1.f;   // 1 / sw
        (void)un2;
#pragma unroll
        for (int ta = 0; ta < 2; ta++)
#pragma unroll
          for (int v = 0; v < 16; v++) {
            const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(t[ta][0][v]), __float_as_uint(t[ta][1][v]), false, false);
            const int c0 = 32 * ta + 8 * (v >> 2) + (v & 3);
            r[c0] = ((__uint_as_float(sw2[0]) * un1) * un1) * unw;
            r[c0 + 4] = ((__uint_as_float(sw2[1]) * un1) * un1) * unw;
          }
        // + XtX (implicit) / lambda_use I (explicit); padded coordinates: identity
        const float* gcol = a.XtX + ln;
#pragma unroll
        for (int m = 0; m < KP; m++) {
          float gv;
          if (m >= k || !lk) gv = (m == ln) ? 1.f : 0.f;
          else if (IMPLICIT) gv = gcol[(size_t)m * k];
          else gv = (m == ln) ? lam_use : 0.f;
          r[m] = (m >= k || !lk) ? gv : r[m] + gv;
        }
      }
    }
    if constexpr (KP != 64) {
      const float* gcol = a.XtX + ln;
#pragma unroll
      for (int m = 0; m < KP; m++) {
        float gv;
        if (m >= k || !lk) gv = (m == ln) ? 1.f : 0.f;   // padded coordinates: identity, rhs 0
        else if (IMPLICIT) gv = gcol[(size_t)m * k];
        else gv = (m == ln) ? lam_use : 0.f;
        r[m] = gv;
      }
    for (int base = p1; base < p2; base += 64) {
      const int ccnt = min(64, p2 - base);
      const int jl = min(lane, ccnt - 1);
      const int idj = a.row_idx[base + jl];
      const float cvj = a.vals[base + jl];
      const float rcj = a.rhs_vals ? a.rhs_vals[base + jl] : cvj;   // coefficient in the right-hand side
      constexpr int PF = 4;                                         // vectors in flight
      float xq[PF];
#pragma unroll
      for (int q = 0; q < PF; q++) {
        const int id = __builtin_amdgcn_readlane(idj, min(q, ccnt - 1));
        xq[q] = lk ? a.X[(size_t)id * k + ln] : 0.f;
      }
      for (int j0 = 0; j0 < ccnt; j0 += PF) {
        float xn[PF];
#pragma unroll
        for (int q = 0; q < PF; q++) {
          const int id = __builtin_amdgcn_readlane(idj, min(j0 + PF + q, ccnt - 1));
          xn[q] = lk ? a.X[(size_t)id * k + ln] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < PF; q++) {
          if (j0 + q < ccnt) {   // wave-uniform
            const float xv = xq[q];
            const float cv = readlane_f(cvj, j0 + q);
            const float f = (IMPLICIT ? cv - 1.f : 1.f) * xv;
            u = fmaf(readlane_f(rcj, j0 + q), xv, u);
            wave_sync();
            if (on) sX[ln] = xv;
            wave_sync();
#pragma unroll
            for (int m4 = 0; m4 < KP / 4; m4++) {
              const float4 b = *reinterpret_cast<const float4*>(sX + 4 * m4);   // broadcast
              r[4 * m4 + 0] = fmaf(b.x, f, r[4 * m4 + 0]);
              r[4 * m4 + 1] = fmaf(b.y, f, r[4 * m4 + 1]);
              r[4 * m4 + 2] = fmaf(b.z, f, r[4 * m4 + 2]);
              r[4 * m4 + 3] = fmaf(b.w, f, r[4 * m4 + 3]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < PF; q++) xq[q] = xn[q];
      }
    }
    }   // (KP != 64)
    if (a.rhs_init && lk) u += a.rhs_init[ln];

    // ---------------- LDL^T, lane i = row i (wrmf_chol_lr.hip's register solve with a guarded pivot) ----------------
    // Pivot step j: r[c] -= l_ij A(j, c) for the columns c > j; A(j, c) = A(c, j) is lane c of register r[j]: its rows of
    // 16 lanes are copied into every row of 16 once per pivot and the multiplier is a DPP row broadcast inside the FMA.  The
    // next pivot column is served first and by v_readlane, so that its chain starts before this pivot's other columns are done.
    // At the end lane i holds, left of the diagonal, row i of L D and, right of it, column i of L d_i: both substitutions
    // read only the lane's own registers and broadcasts; the forward substitution rides along.
    bool bad = false;
    float dinv = 1.f;
    float pj = readlane_f(r[0], 0);
    cw_sfor<KP>([&](auto jt) {
      constexpr int j = decltype(jt)::value;
      bad = bad || !(pj > 0.f);
      const float i0 = __builtin_amdgcn_rcpf(pj);
      const float inv = fmaf(fmaf(-pj, i0, 1.f), i0, i0);
      const float uj = readlane_f(u, j);
      if (ln == j) dinv = inv;
      const float lij = ln > j ? r[j] * inv : 0.f;   // L_ij; rows <= j are finished
      u = fmaf(-lij, uj, u);
      if constexpr (j + 1 < KP) {
        r[j + 1] = fmaf(-lij, readlane_f(r[j], j + 1), r[j + 1]);
        pj = readlane_f(r[j + 1], j + 1);
        if constexpr (j + 2 < KP) {
          float rep[4];
          dpp_ready(r[j]);
          rows_to_all<(KP > 32 ? 4 : 2)>(r[j], rep);
          dpp_ready(rep[0], rep[1], rep[2], rep[3]);
          cw_sfor<KP - j - 2>([&](auto ct) {
            constexpr int c = j + 2 + decltype(ct)::value;
            fnma_row_bcast<c % 16>(r[c], rep[c / 16], lij);
          });
        }
      }
    });
    // backward: z_c = (u_c - sum_{c' > c} d_c L[c'][c] z_c') / d_c, largest index first
    float acc = 0.f, z = 0.f;
#pragma unroll
    for (int c = KP - 1; c >= 0; c--) {
      if (ln == c) z = (u - acc) * dinv;
      const float zc = readlane_f(z, c);
      acc = fmaf(ln < c ? r[c] : 0.f, zc, acc);
    }
    if (bad) {   // wave-uniform: the general solver re-solves the row and owns its loss term (wrmf_lu.hip)
      int pos = 0;
      if (lane == 0) pos = atomicAdd(a.fail_counter, 1);
      pos = rfl(pos);
      if (pos < a.fail_cap) {
        if (lane == 0) a.fail_rows[pos] = row;
      } else if (lk) {
        yrow[ln] = 0.f;   // no room in the list: unresolved, zeroed like a singular row
      }
      continue;
    }
    if (lk) yrow[ln] = z;

    // ---------------- loss row term: lane j takes non-zero j of a chunk ----------------
    wave_sync();
    if (on) sX[ln] = z;
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
    const float xxp = wave_sum(z * z);
    wloss += IMPLICIT ? (double)lpart + a.lambda_loss * (double)xxp : (double)(lpart + lam_use * xxp);
  }
  if (lane == 0) a.loss_partials[loss_slot0 + blockIdx.x] = wloss;
}

}  // namespace

bool chol_wave_supported(int k) { return padded_rank(k) == 32 || padded_rank(k) == 64; }

// grid workgroups of one wave (a row each at a time); loss partials [loss_slot0, loss_slot0 + grid)
hipError_t launch_als_chol_wave(const AlsArgs& a, bool implicit, int grid, int loss_slot0, hipStream_t s, hipEvent_t* ev_slot) {
  const int KP = padded_rank(a.k);
#define RSP_CW(KPV, IMP)                                                                  \
  {                                                                                       \
    auto kern = als_chol_wave_kernel<KPV, IMP>;                                           \
    prof_note(ev_slot, reinterpret_cast<const void*>(kern));                              \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, s, a, loss_slot0);                  \
    return hipGetLastError();                                                             \
  }
  if (KP == 32) {
    if (implicit) RSP_CW(32, true) else RSP_CW(32, false)
  }
  if (KP == 64) {
    if (!a.wave_stats) return hipErrorInvalidValue;   // (the operand scales of the matrix-core assembly)
    if (implicit) RSP_CW(64, true) else RSP_CW(64, false)
  }
#undef RSP_CW
  return hipErrorInvalidValue;
}

}  // namespace rsparse_hip
