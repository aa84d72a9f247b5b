// Device-side building blocks shared by the ALS kernels (gfx950 / wave64 only).
//
// Lane <-> data mappings used throughout (KP = rank padded to 32/64/128, T = tile capacity in
// non-zeros, one tile row = one gathered factor vector, row stride LDT = KP + 4 floats):
//
//   "k-mode"    lane l owns vector elements [l*EPL, l*EPL+EPL), EPL = KP/64 (1 for KP <= 64):
//               CG state x, r, p, Ap live in registers this way; u = X_nnz * w is an AXPY over
//               tile rows (contiguous ds_read_b64 per lane, conflict free).
//   "nnz-mode"  lane l owns non-zero j = l % T and the k-range split h = l / T:
//               t = X_nnz^T v is a per-lane dot product down a tile row (ds_read_b128; the +4 pad
//               makes the 16-lane service groups hit distinct banks) followed by a log2(64/T)-step
//               cross-split shuffle.  Lane j then holds t_j, c_j, w_j.
//
// The two GEMVs of every CG step therefore read the same LDS tile in its two orientations and
// nothing is transposed or re-gathered.
#pragma once
#include <hip/hip_runtime.h>

namespace rsparse_hip {
namespace dev {

__device__ __forceinline__ void wave_sync() {
  // LDS traffic of one wave is executed in order; this only pins the compiler's ordering.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float readlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// acc -= u(lane E of the row of 16 lanes this lane sits in) * l: v_fmac_f32 with a DPP row broadcast on its first operand.
// 4.8 cycles per wave against 8.4 + 4.3 for v_readlane + v_fma (tools/probes/valu_rate_probe.hip).  Needs EXEC = all ones.
template <int E>
__device__ __forceinline__ void fnma_row_bcast(float& acc, const float u, const float l) {
  asm("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(u), "v"(l), "n"(E));
}
// A register that a DPP operand (or a lane swap) is about to read must not have been written by the one or two vector
// instructions before it (2 wait states).  hipcc inserts them between instructions it knows; it does not look inside inline
// asm, neither as the writer nor as the reader -- these tie an s_nop to the registers (the asm "rewrites" them, so the real
// writers stay in front of it and the readers behind).
__device__ __forceinline__ void dpp_ready(float& a) { asm("s_nop 1" : "+v"(a)); }
__device__ __forceinline__ void dpp_ready(float& a, float& b) { asm("s_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void dpp_ready(float& a, float& b, float& c, float& d) {
  asm("s_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
// rep[g] = row g (lanes 16 g .. 16 g + 15) of v in every row of 16 lanes, g < NG (two or three lane-swap instructions)
template <int NG>
__device__ __forceinline__ void rows_to_all(const float v, float (&rep)[4]) {
  const unsigned u = __float_as_uint(v);
  const auto h = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // h[0] = rows (0, 1, 0, 1), h[1] = rows (2, 3, 2, 3)
  const auto lo = __builtin_amdgcn_permlane16_swap(h[0], h[0], false, false);   // rows (0, 0, 0, 0), (1, 1, 1, 1)
  rep[0] = __uint_as_float(lo[0]);
  rep[1] = __uint_as_float(lo[1]);
  if constexpr (NG > 2) {
    const auto hi = __builtin_amdgcn_permlane16_swap(h[1], h[1], false, false);
    rep[2] = __uint_as_float(hi[0]);
    rep[3] = __uint_as_float(hi[1]);
  } else {
    rep[2] = rep[3] = 0.f;
  }
}

// Sum over the 64 lanes, result uniform.  Needs EXEC = all ones.  Fixed order -> deterministic.
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp<0xB1>(v);   // quad_perm:[1,0,3,2]
  v += dpp<0x4E>(v);   // quad_perm:[2,3,0,1]
  v += dpp<0x141>(v);  // row_half_mirror
  v += dpp<0x140>(v);  // row_mirror  -> every lane of a 16-lane row holds the row sum
  const float s0 = readlane_f(v, 0), s1 = readlane_f(v, 16);
  const float s2 = readlane_f(v, 32), s3 = readlane_f(v, 48);
  return (s0 + s1) + (s2 + s3);
}

template <int KP>
struct Geo {
  static constexpr int EPL = KP >= 64 ? KP / 64 : 1;
  static constexpr int LDT = KP + 4;
};

// t_j = sum_k tile[j][k] * vec[k]; returned in every lane whose (lane % T) == j.
template <int KP, int T>
__device__ __forceinline__ float tile_dot(const float* tile, const float* vec, int lane) {
  constexpr int S = 64 / T, HK = KP / S, LDT = Geo<KP>::LDT;
  static_assert(HK % 4 == 0, "k-range per split must be a multiple of 4");
  const int j = lane % T, h = lane / T;
  const float4* a = reinterpret_cast<const float4*>(tile + j * LDT + h * HK);
  const float4* b = reinterpret_cast<const float4*>(vec + h * HK);
  float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
  for (int q = 0; q < HK / 4; q++) {
    const float4 av = a[q], bv = b[q];
    t0 = fmaf(av.x, bv.x, t0);
    t1 = fmaf(av.y, bv.y, t1);
    t2 = fmaf(av.z, bv.z, t2);
    t3 = fmaf(av.w, bv.w, t3);
  }
  float t = (t0 + t1) + (t2 + t3);
#pragma unroll
  for (int m = T; m < 64; m <<= 1) t += __shfl_xor(t, m);
  return t;
}

// acc += sum_{j<cnt} w_j * tile[j][:]   (k-mode); w_j is read from lane j.
template <int KP>
__device__ __forceinline__ void tile_axpy(const float* tile, float w, int cnt, int lane, bool active,
                                          float (&acc)[Geo<KP>::EPL]) {
  constexpr int EPL = Geo<KP>::EPL, LDT = Geo<KP>::LDT;
  const float* col = tile + lane * EPL;
  auto step = [&](int j) {
    const float wj = readlane_f(w, j);
    if constexpr (EPL == 2) {
      const float2 xv = *reinterpret_cast<const float2*>(col + j * LDT);
      acc[0] = fmaf(wj, xv.x, acc[0]);
      acc[1] = fmaf(wj, xv.y, acc[1]);
    } else {
      const float xv = active ? col[j * LDT] : 0.f;
      acc[0] = fmaf(wj, xv, acc[0]);
    }
  };
  int j = 0;
  for (; j + 4 <= cnt; j += 4) {  // hand-unrolled: the readlane keeps the compiler from doing it
    step(j);
    step(j + 1);
    step(j + 2);
    step(j + 3);
  }
  for (; j < cnt; j++) step(j);
}

// acc += sign * sum_{kk in [kk0,kk1)} G[kk][:] * vec[kk]   (G symmetric KP x KP in LDS, k-mode).
template <int KP>
__device__ __forceinline__ void gram_mv(const float* sG, const float* vec, int kk0, int kk1, int lane,
                                        bool active, float sign, float (&acc)[Geo<KP>::EPL]) {
  constexpr int EPL = Geo<KP>::EPL;
  float part[EPL];
#pragma unroll
  for (int u = 0; u < EPL; u++) part[u] = 0.f;
  const float* col = sG + lane * EPL;
#pragma unroll 2
  for (int kk = kk0; kk < kk1; kk += 4) {
    const float4 vb = *reinterpret_cast<const float4*>(vec + kk);
    const float vv[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
    for (int u4 = 0; u4 < 4; u4++) {
      if constexpr (EPL == 2) {
        const float2 g = *reinterpret_cast<const float2*>(col + (kk + u4) * KP);
        part[0] = fmaf(vv[u4], g.x, part[0]);
        part[1] = fmaf(vv[u4], g.y, part[1]);
      } else {
        const float g = active ? col[(kk + u4) * KP] : 0.f;
        part[0] = fmaf(vv[u4], g, part[0]);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < EPL; u++) acc[u] = fmaf(sign, part[u], acc[u]);
}

template <int KP>
__device__ __forceinline__ void put_vec(float* vec, const float (&v)[Geo<KP>::EPL], int lane,
                                        bool active) {
  constexpr int EPL = Geo<KP>::EPL;
  if constexpr (EPL == 2) {
    *reinterpret_cast<float2*>(vec + lane * 2) = make_float2(v[0], v[1]);
  } else {
    if (active) vec[lane] = v[0];
  }
}

// Gather `cnt` (<= T) factor vectors X[:, idx_j] into tile rows 0..cnt-1.  idx_j is held by lane j.
// VEC: rank % 4 == 0 and X 16-byte aligned -> 16 B per lane, 64*16/(4*KP) vectors per instruction.
// Columns [k, KP) of the tile are never written with non-zeros (they were zeroed at kernel start).
template <int KP, int T, bool VEC>
__device__ __forceinline__ void gather_chunk(const float* __restrict__ X, int k, int myidx, int cnt,
                                             float* tile, int lane) {
  constexpr int LDT = Geo<KP>::LDT;
  if constexpr (VEC) {
    constexpr int LPV = KP / 4, VPI = 64 / LPV, NLD = T / VPI;
    const int c4 = lane % LPV, jo = lane / LPV;
    float4 buf[NLD];
#pragma unroll
    for (int q = 0; q < NLD; q++) {
      const int j = q * VPI + jo;
      const int id = __shfl(myidx, j);
      buf[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q * VPI < cnt) {  // wave-uniform
        if (j < cnt && c4 * 4 < k)
          buf[q] = *reinterpret_cast<const float4*>(X + (size_t)id * k + c4 * 4);
      }
    }
#pragma unroll
    for (int q = 0; q < NLD; q++) {
      const int j = q * VPI + jo;
      if (q * VPI < cnt) {
        if (j < cnt) *reinterpret_cast<float4*>(tile + j * LDT + c4 * 4) = buf[q];
      }
    }
  } else {
    for (int j = 0; j < cnt; j++) {
      const int id = __builtin_amdgcn_readlane(myidx, j);
      const float* src = X + (size_t)id * k;
      for (int e = lane; e < k; e += 64) tile[j * LDT + e] = src[e];
    }
  }
}

}  // namespace dev
}  // namespace rsparse_hip
