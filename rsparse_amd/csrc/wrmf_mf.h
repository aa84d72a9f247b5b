// Device helpers shared by the wave-per-row kernels that keep their matrix-core accumulators in the accumulator file BY NAME
// (wrmf_chol_mf.hip: the exact solve at rank 65..128; wrmf_cg_mf.hip: the long rows of the conjugate-gradient half-iteration).
// The including file defines, before the include, MF_A0 (first accumulator register of tile 0: `constexpr int`) and MF_TOP (string:
// the highest accumulator register the kernel names, listed as a clobber so that the kernel descriptor allocates the range).
// Why the registers are named, what hipcc must be kept from doing and how the build checks it: wrmf_chol_mf.hip's header,
// DESIGN.md 3.3, tools/dbg/acc_audit.py.
#pragma once
#include <utility>

#include "wrmf_device.h"

namespace rsparse_hip {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <class F, int... I>
__device__ __forceinline__ void mf_sfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void mf_sfor(F&& f) {
  mf_sfor_impl(f, std::make_integer_sequence<int, N>{});
}

// x (already scaled into fp16's range) -> fl16(x), fl16(x - fl16(x)) for a pair; the residual is exact in fp32
__device__ __forceinline__ void mf_split(const float x0, const float x1, unsigned& hi, unsigned& lo) {
  // (scalar subtractions on purpose: a packed fp32 instruction -- v_pk_add_f32 and friends -- issued behind a matrix instruction
  // waits ~28 cycles for it where a plain one is free, tools/probes/mfma_filler_probe.hip; these files are also compiled with
  // -fno-slp-vectorize so that hipcc does not pair them up again)
  const f32x2 v = {x0, x1};
  const f16x2 h = __builtin_convertvector(v, f16x2);
  const float r0 = x0 - (float)h[0], r1 = x1 - (float)h[1];
  const f32x2 r = {r0, r1};
  const f16x2 l = __builtin_convertvector(r, f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ f16x8 mf_pack(const unsigned a, const unsigned b, const unsigned c, const unsigned d) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = {a, b, c, d};
  return __builtin_bit_cast(f16x8, v);
}
// biased exponent e of the power of two that brings `vmax` into [2^13, 2^14); 2^(e - 127) is the scale
__device__ __forceinline__ int mf_scale_exp(float vmax) {
  const int eb = (int)((__float_as_uint(vmax) >> 23) & 0xffu);
  return min(253, max(1, 267 - eb));
}
__device__ __forceinline__ float mf_pow2(int biased) { return __uint_as_float((unsigned)biased << 23); }

// sixteen fp32 registers of one coordinate half (register s = non-zero s of the step, lane l = coordinate l of the half) ->
// the fp16-term operands of the two 32-coordinate blocks of that half: lanes (n, 0) get the non-zeros 0..7, lanes (n, 1) the
// non-zeros 8..15 (one lane swap per register pair, as in wrmf_chol_wave.hip)
__device__ __forceinline__ void mf_operands(const float (&x)[16], f16x8& h0, f16x8& l0, f16x8& h1, f16x8& l1) {
  float b0[8], b1[8];
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[e]), __float_as_uint(x[8 + e]), false, false);
    b0[e] = __uint_as_float(sw2[0]);
    b1[e] = __uint_as_float(sw2[1]);
  }
  unsigned hh0[4], ll0[4], hh1[4], ll1[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    mf_split(b0[2 * q], b0[2 * q + 1], hh0[q], ll0[q]);
    mf_split(b1[2 * q], b1[2 * q + 1], hh1[q], ll1[q]);
  }
  h0 = mf_pack(hh0[0], hh0[1], hh0[2], hh0[3]); l0 = mf_pack(ll0[0], ll0[1], ll0[2], ll0[3]);
  h1 = mf_pack(hh1[0], hh1[1], hh1[2], hh1[3]); l1 = mf_pack(ll1[0], ll1[1], ll1[2], ll1[3]);
}

constexpr int mf_tid(int I, int K) { return I * (I + 1) / 2 + K; }   // the ten lower tiles, I >= K; tile t = a[MF_A0 + 16 t : MF_A0 + 16 t + 15]

// ---- the accumulator file by name (see the header) ----
template <int R>
__device__ __forceinline__ float mf_rd() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(x) : "i"(MF_A0 + R));
  return x;
}
template <int R>
__device__ __forceinline__ void mf_wr(const float x) {
  asm volatile("v_accvgpr_write_b32 a%c1, %0" ::"v"(x), "i"(MF_A0 + R) : MF_TOP);   // (the clobber makes the kernel own the range)
}
// the matrix pipe has drained: results of the last (<= 16-pass) matrix instruction may be read, its registers written
#define MF_DRAIN() asm volatile("s_nop 15\n\ts_nop 7" ::: "memory")
// tile T += A B^T, 16 non-zeros (fp16 terms); opens with the wait states of a VALU-written operand
template <int T>
__device__ __forceinline__ void mf_mma16(const f16x8& a, const f16x8& b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "i"(MF_A0 + 16 * T), "i"(MF_A0 + 16 * T + 15));
}
// tile T += a b^T, two panel columns (exact fp32 products)
template <int T>
__device__ __forceinline__ void mf_mma2(const float a, const float b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "i"(MF_A0 + 16 * T), "i"(MF_A0 + 16 * T + 15));
}

// One element of a gathered vector, NOT waited for: hipcc follows every plain load whose value passes through a select with
// s_waitcnt vmcnt(0) and sinks it to its use (32 serial round trips per step in the first version of this kernel).
// base: the vector (uniform, scalar registers), voff: this lane's byte offset into it.
__device__ __forceinline__ void mf_ld(float& d, const float* base, const int voff) {
  asm volatile("global_load_dword %0, %1, %2" : "=&v"(d) : "v"(voff), "s"(base));
}
// ... and the one wait in front of their first use (tied to the registers, 8 at a time: an asm statement takes 30 operands)
__device__ __forceinline__ void mf_tie8(float* x) {
  asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
}
__device__ __forceinline__ void mf_wait(float (&a)[16], float (&b)[16]) {
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
               :: "memory");
  mf_tie8(a + 8);
  mf_tie8(b);
  mf_tie8(b + 8);
}


}  // namespace
}  // namespace rsparse_hip
