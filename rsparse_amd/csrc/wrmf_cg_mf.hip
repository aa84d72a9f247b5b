// Conjugate-gradient half-iteration, rows of 513 .. kCgMfMax non-zeros at rank 128, implicit feedback: one WAVE per row, one pass
// over the row, the normal equations in the wave's matrix-core ACCUMULATOR registers (gfx950, wave64; round 6).
//
// Replaces, for those rows, the streaming part of als_ne_kernel (wrmf_ne.hip) -- cg_solver_implicit<T> on the row's k x k system
// (inst/include/wrmf_implicit.hpp:8-32, the same operator evaluated from the assembled matrix) and the loss term from the
// accumulators:
//     M1 = X_nnz diag(c - 1) X_nnz^T,   M2 = X_nnz X_nnz^T,   b = X_nnz c,   sum c
//     A = XtX + M1;   y = CG(A, b; warm start, cg_steps);   sum_j c_j (1 - x_j.y)^2 = sum c - 2 y.b + y^T (M1 + M2) y.
// als_ne_kernel moves these rows' bytes at 0.42 of the HBM peak (20 ms per launch on the bench line, the dominant launch of
// rounds 3-5): four waves consume the same 16-non-zero step from an LDS ring, and 114..227 of a step's 315..500 instructions are
// scalar bookkeeping of that ring (DESIGN.md 3.2 "Round 5").  Here a wave has no partner: the row's vectors arrive by LDS-DMA in
// a ring of its own, the accumulator tiles are the wave's.  What conjugate gradient needs next to the system matrix is the loss-only
// matrix M2: 20 tiles = 320 accumulator registers.  One wave per SIMD can hold them -- the accumulator file a0..a255 takes M1
// (tiles 0..9) and six tiles of M2, the other four tiles of M2 are vector registers of the compiler's (a plain accumulate chain:
// nothing hipcc can get wrong) -- and one wave per SIMD issues strictly in order: a step's vector work and its 40 matrix
// instructions add up unless the instruction stream interleaves them itself (the pipelined step below; what a vector instruction
// costs behind a matrix instruction: tools/probes/mfma_filler_probe.hip).  15.0 ms per launch at 0.54 of the HBM peak, at the
// chip's power limit (1.57 GHz: DESIGN.md 3.2 "Round 6").
//
// Per row: stream (M1 from two fp16 terms of 2^e sqrt(c - 1) x, three products; M2 from the leading fp16 term of 2^e x alone, one
// product: it feeds only the loss, as in wrmf_ne.hip) -> unscale -> conjugate gradient with A p evaluated FROM THE TILES: tile
// (I, K) of the lower triangle holds, at lane (n, hf), register v, the entry [(block I, row n)][(block K, row rho(v, hf))] -- row r of
// block I being coordinate 4 r + I, see "COORDINATES" below --, so
//     (A p)_I += sum over the lane's registers of  tile * p[column]          ("direct": 16 FMAs per tile, then the halves added)
//     (A p)_K += sum over the LANES of             tile * p[row]   (I > K)   ("transposed": through a 32 x 36 LDS tile)
// with XtX's tiles added on the fly from a copy in LDS in the same lane / register order (shared by the workgroup's four waves,
// which otherwise know nothing of each other).  The loss reads y^T M1 y and y^T M2 y off the tiles ("direct" part only).
// Double scalars rsold / alpha / beta as the reference holds them (:18); the 1e-10 exit (:27).
#include <cstdio>
#include <utility>

#pragma clang diagnostic ignored "-Winline-asm"   // (the named accumulator registers are "reserved": that is the point)

#include "wrmf_internal.h"
#include "wrmf_device.h"

namespace rsparse_hip {
namespace {
using namespace dev;
// the accumulator file is this kernel's alone (forced attribute amdgpu-agpr-alloc=0, see wrmf_chol_mf.hip): a0..a159 = M1's ten tiles
// (then A = XtX + M1 is NOT formed in them: M1 stays pure for the loss), a160..a255 = tiles 0..5 of M2
constexpr int MF_A0 = 0;
#define MF_TOP "a255"
}  // namespace
}  // namespace rsparse_hip

#include "wrmf_mf.h"

namespace rsparse_hip {
namespace {

constexpr float kCgTolMf = 1e-10f;   // CG_TOL, inst/include/wrmf.hpp:22
// Measured and not kept, first version (two REGISTER buffers, asm loads, one `s_waitcnt vmcnt(0)` per step: 17.3 ms per launch):
//  - the previous step's matrix instructions in five groups of eight BETWEEN the pieces of this step's vector work, their operands
//    waiting in LDS: 19.8 ms (profiles/r06/r6o_*) -- the staging writes, 60 operand reads per step and the lost scheduling freedom
//    cost more than the 1280 cycles of matrix-pipe time they were meant to hide;
//  - one extra load per step touching the 64 cache lines of step s + 2 so that they are in L2 early: 17.7 ms (profiles/r06/r6p_*);
//  - a third register buffer: 96 + 64 (M2's last tiles) + a step's operands do not fit 256 registers (31 scratch accesses per step).
// the leading fp16 term alone (M2: one product)
__device__ __forceinline__ void cgm_operands_hi(const float (&x)[16], f16x8& h0, f16x8& h1) {
  float b0[8], b1[8];
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[e]), __float_as_uint(x[8 + e]), false, false);
    b0[e] = __uint_as_float(sw2[0]);
    b1[e] = __uint_as_float(sw2[1]);
  }
  unsigned hh0[4], hh1[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const f32x2 v0 = {b0[2 * q], b0[2 * q + 1]}, v1 = {b1[2 * q], b1[2 * q + 1]};
    hh0[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(v0, f16x2));
    hh1[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(v1, f16x2));
  }
  h0 = mf_pack(hh0[0], hh0[1], hh0[2], hh0[3]);
  h1 = mf_pack(hh1[0], hh1[1], hh1[2], hh1[3]);
}

// A step's fp16 operands by 32-coordinate block: M1's two terms and M2's one
struct CgmOps {
  f16x8 ah[4], al[4], mh[4];
};
// The 40 matrix instructions of a step in the order the pipelined loop issues them BETWEEN the next step's vector work: block
// column K = 0 first (16: its operands die first), then K = 1 (12), then K = 2 and 3 together (12); inside a group product-major
// (kind 0: M2's one product, 1..3: M1's hi hi / hi lo / lo hi), so consecutive instructions write different tiles.
struct CgmProd {
  int T, kind, K, I;
};
constexpr CgmProd cgm_prod(int idx) {
  if (idx < 16) return {mf_tid(idx % 4, 0), idx / 4, 0, idx % 4};
  idx -= 16;
  if (idx < 12) return {mf_tid(1 + idx % 3, 1), idx / 3, 1, 1 + idx % 3};
  idx -= 12;
  const int w = idx % 3, K = w == 2 ? 3 : 2, I = w == 0 ? 2 : 3;
  return {mf_tid(I, K), idx / 3, K, I};
}
// tile T += A B^T without the operand wait states of mf_mma16: the operands were written a step ago
template <int T>
__device__ __forceinline__ void cgm_mma16(const f16x8& a, const f16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "i"(MF_A0 + 16 * T), "i"(MF_A0 + 16 * T + 15));
}

template <int NOPS>
struct CgmWave {               // per wave
  union {
    float ring[3][2048];       // streaming: three steps of 16 vectors, written by LDS-DMA: non-zero j = 8 h + i of a step at [i][h][128]
    float T[32 * 36];          // solve: the transposed part of a product: [column][lane], rows padded to 36 floats (conflict-free b128 reads)
  };
  int midx[3][64];             // three chunks of 64 (index, confidence) pairs, by LDS-DMA too
  float mval[3][64];
  float gval[3][64];           // (SYM) M1's operand scale per non-zero: sqrt((c - 1) 2^e')
  unsigned long long maddr[3][64];   // the non-zero's vector: its address, less the offset field of the LDS-DMA piece that will fetch it
  float vec[128];              // a vector in "register = column" order's source: p (or y) by coordinate
  float x0[128];               // warm start
  float rhs[128];              // b
  double loss;
};
template <int NOPS>
struct CgmSmem {
  float4 G[10][64][4];         // XtX's lower tiles in the accumulator's lane / register order: [tile][lane][4 x float4 = 16 registers]
  CgmWave<NOPS> w[4];
};

// y_I(n) += sum_v tile[v] * pc[K][v] over the lane's registers, for every lower tile; off-diagonal tiles also feed the transposed
// part.  RD(t, v) reads element v of tile t.
template <class RD, class GV, class WV>
__device__ __forceinline__ void cgm_matvec(RD&& rd, GV&& gtile, WV& sw, const int n, const int hf, const int ln,
                                           const float (&pr)[4], float (&out)[4]) {
  // pr[I] = p[32 I + n] (lane = row).  The column-ordered copy: pc[K][v] = p[32 K + rho(v, hf)] from LDS
  wave_sync();
  if (hf == 0) {
#pragma unroll
    for (int I = 0; I < 4; I++) sw.vec[32 * I + n] = pr[I];
  }
  wave_sync();
  float dsum[4] = {0.f, 0.f, 0.f, 0.f};
  mf_sfor<4>([&](auto kt) {
    constexpr int K = decltype(kt)::value;
    float pc[16];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const float4 t = *reinterpret_cast<const float4*>(&sw.vec[32 * K + 8 * q + 4 * hf]);
      pc[4 * q] = t.x; pc[4 * q + 1] = t.y; pc[4 * q + 2] = t.z; pc[4 * q + 3] = t.w;
    }
    float ts[16];   // transposed partials of block column K: sum over the tiles (I, K), I > K, of tile[v] * p[32 I + n]
#pragma unroll
    for (int v = 0; v < 16; v++) ts[v] = 0.f;
    mf_sfor<4 - K>([&](auto st) {
      constexpr int I = K + decltype(st)::value;
      constexpr int T = mf_tid(I, K);
      float g[16];
      gtile(std::integral_constant<int, T>{}, g);
      mf_sfor<16>([&](auto vt) {
        constexpr int v = decltype(vt)::value;
        const float av = rd(std::integral_constant<int, T>{}, std::integral_constant<int, v>{}) + g[v];
        dsum[I] = fmaf(av, pc[v], dsum[I]);
        if constexpr (I > K) ts[v] = fmaf(av, pr[I], ts[v]);
      });
    });
    if constexpr (K < 3) {
      // sum ts[v] over the 32 lanes of each half: through LDS, column rho(v, hf) = row of a 32 x 36 tile
      wave_sync();
#pragma unroll
      for (int v = 0; v < 16; v++) sw.T[(8 * (v >> 2) + 4 * hf + (v & 3)) * 36 + n] = ts[v];
      wave_sync();
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const float4 t = *reinterpret_cast<const float4*>(&sw.T[n * 36 + 4 * q]);
        s0 += t.x + t.z;
        s1 += t.y + t.w;
      }
      out[K] = s0 + s1;   // (completed below with the direct part)
    } else {
      out[K] = 0.f;
    }
  });
  // the direct part: the two halves of the wave hold the two halves of a row's columns
#pragma unroll
  for (int I = 0; I < 4; I++) {
    const unsigned du = __float_as_uint(dsum[I]);
    const auto sw2 = __builtin_amdgcn_permlane32_swap(du, du, false, false);   // [0]: the lanes (n, 0)'s value everywhere, [1]: (n, 1)'s
    out[I] += __uint_as_float(sw2[0]) + __uint_as_float(sw2[1]);
  }
}

// sum over a vector held lane = row (four registers, both halves of the wave hold the same values)
__device__ __forceinline__ float cgm_dot(const float (&a)[4], const float (&b)[4], const int hf) {
  float s = (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
  return wave_sum(hf == 0 ? s : 0.f);
}

template <bool SYM>
__device__ __forceinline__ void als_cg_mf_body(const AlsArgs& a, const int32_t* __restrict__ rows, const int n_rows, const int loss_slot0) {
  extern __shared__ __attribute__((aligned(16))) char cgm_smem[];   // (more than the 64 KB a static array may take)
  CgmSmem<SYM ? 12 : 20>& sm = *reinterpret_cast<CgmSmem<SYM ? 12 : 20>*>(cgm_smem);
  const int lane = threadIdx.x & 63, wv = rfl((int)(threadIdx.x >> 6));
  CgmWave<SYM ? 12 : 20>& sw = sm.w[wv];
  constexpr int k = 128;
  {
    // two instantiations are launched; "some confidence < 1" (word 2 of the values scan) says which one works
    const bool below_one = a.ne_stats[2] != 0u;
    if (below_one == SYM) {
      if (lane == 0) a.loss_partials[loss_slot0 + 4 * blockIdx.x + wv] = 0.0;
      return;
    }
  }
  // COORDINATES.  Row n of coordinate block I is coordinate 4 n + I (not 32 I + n): a lane then reads its four blocks' entries of a
  // vector with ONE ds_read_b128 from the ring (the matrix pipe does not care which coordinate a tile row is; XtX's tiles, the
  // warm start and the stored row follow the same map, everything between them lives in (block, row) order).
  // XtX's lower tiles -> LDS in the accumulator's order: tile (I, K), lane (n, hf), register v = XtX[(K, rho(v, hf))][(I, n)]
  for (int e = threadIdx.x; e < 10 * 64 * 16; e += 256) {
    const int t = e / 1024, l = (e >> 4) & 63, v = e & 15;
    const int I = t >= 6 ? 3 : (t >= 3 ? 2 : (t >= 1 ? 1 : 0)), K = t - I * (I + 1) / 2;
    const int nn = l & 31, hh = l >> 5;
    reinterpret_cast<float*>(&sm.G[t][l][0])[v] = a.XtX[(size_t)(4 * (8 * (v >> 2) + 4 * hh + (v & 3)) + K) * k + 4 * nn + I];
  }
  if (lane == 0) sw.loss = 0.0;
  __syncthreads();

#ifdef RSP_MF_PROF   // dev builds (tools/gpu_cgmf_prof.sh): s_memtime ticks per phase, summed over the waves into a.ne_prof[16 ..]
  unsigned long long pt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pt0 = __builtin_amdgcn_s_memtime();
  const unsigned long long pt_start = pt0, rt_start = __builtin_amdgcn_s_memrealtime();   // (100 MHz: ticks / realtime = the shader clock)
#define CGM_TICK(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pt[i] += t_ - pt0; pt0 = t_; }
#define CGM_COUNT(i, n) pt[i] += (n);
#else
#define CGM_TICK(i)
#define CGM_COUNT(i, n)
#endif
  auto uni = [](const float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); };
  const int ex = rfl(mf_scale_exp(fmaxf(__uint_as_float(a.ne_stats[0]), 1e-30f)));
  const float wmax = uni(fmaxf(__uint_as_float(a.ne_stats[1]) - 1.f, 1.f));
  const int ewb = rfl((int)((__float_as_uint(wmax) >> 23) & 0xffu));
  const float sx = uni(mf_pow2(ex)), sw_ = uni(mf_pow2(min(253, max(1, 253 - ewb))));   // sw_ = 2^(126 - ewb) <= 1 / wmax
  const float un1 = uni(mf_pow2(254 - ex));                                              // 1 / sx
  const float unw = uni(mf_pow2(254 - min(253, max(1, 253 - ewb))));                     // 1 / sw_
  const int n_waves = 4 * gridDim.x;

  for (int it = 4 * blockIdx.x + wv; it < n_rows; it += n_waves) {
    CGM_TICK(7)
    const int row = rfl(rows[it]);
    const int p1 = rfl(a.col_ptrs[row]), p2 = rfl(a.col_ptrs[row + 1]);
    float* yrow = a.Y + (size_t)row * k;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    // warm start and the accumulators
    sw.x0[ln] = yrow[ln];
    sw.x0[64 + ln] = yrow[64 + ln];
    MF_DRAIN();
    mf_sfor<256>([&](auto rt) { mf_wr<decltype(rt)::value>(0.f); });
    f32x16 hi[4];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int e = 0; e < 16; e++) hi[t][e] = 0.f;
    float u0 = 0.f, u1 = 0.f, csum = 0.f;   // rhs (lane = coordinate), sum of the confidences (lane j: its slots')
    float ua[4] = {0.f, 0.f, 0.f, 0.f};   // SYM: rhs partials, row n of block I over this half's non-zeros
    {
      // A step = 16 non-zeros = 8 KB of vectors.  They arrive by LDS-DMA (global_load_lds_dwordx4: a lane brings 16 bytes, an
      // instruction two whole vectors -- lanes 0..31 non-zero i, lanes 32..63 non-zero 8 + i -- into 1 KB of LDS at M0) in a ring of
      // three steps per wave, and so do the row's indices and confidences, 64 at a time (a "chunk" = four steps).  Nothing in flight
      // has a register: steps s + 1 and s + 2 are on their way while step s is consumed (twice what the two register buffers of the
      // first version held, which moved 3.7 TB/s: 8 KB per wave x 4 waves per CU in flight is what the memory system returned in
      // ~2 us), the queue is never drained (counted waits: the loads return in order), and the 32 registers of the second buffer are
      // free.  Per sub-step: wait until step s has landed; request step s + 2 (its slot was consumed in sub-step s - 1) and, every
      // fourth sub-step, chunk c + 2 of the indices; read step s out of LDS (lane = coordinate), right-hand side, split, 40 matrix
      // instructions.
      const int nsteps = (p2 - p1 + 15) >> 4;
      const int hfl = ln >> 5;
      // a lane's 16 bytes of the vector its half of the wave fetches, from the first vector on (the address table holds offsets)
      const unsigned long long xoff = reinterpret_cast<unsigned long long>(a.X) + 16ull * (unsigned)(ln & 31);
      const unsigned ring_lds = (unsigned)(uintptr_t)&sw.ring[0][0];   // (wave-uniform: wv went through readfirstlane)
      const unsigned meta_lds = (unsigned)(uintptr_t)&sw.midx[0][0];
      // chunk c of the row's (index, confidence) pairs -> sw.midx / sw.mval [c % 3]; positions beyond the row repeat its last entry
      auto request_meta = [&](const int c) __attribute__((always_inline)) {
        const int pos = min(p1 + 64 * c + ln, p2 - 1);
        const int* gi = a.row_idx + pos;
        const float* gv = a.vals + pos;
        const unsigned dst = meta_lds + 256u * (unsigned)(c % 3);
        unsigned keep;
        asm volatile(
            "s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[l]\n\ts_nop 0\n\t"
            "global_load_lds_dword %[gi], off\n\ts_add_u32 m0, m0, 0x300\n\ts_nop 0\n\t"
            "global_load_lds_dword %[gv], off\n\t"
            "s_mov_b32 m0, %[k]"
            : [k] "=&s"(keep)
            : [l] "s"(dst), [gi] "v"(gi), [gv] "v"(gv)
            : "memory", "scc");
      };
      // the 16 vectors of step st -> sw.ring[st % 3]: non-zero j = 8 h + i at [i][h][128].  ONE write of M0 (the middle of the slot):
      // a piece's offset field (-4096 .. 3072) moves its LDS destination AND its global address, and the chunk's address table
      // has the latter taken out again.
      auto request = [&](const int st, const int slot) __attribute__((always_inline)) {
        const uint4* ap = reinterpret_cast<const uint4*>(&sw.maddr[(st >> 2) % 3][16 * (st & 3) + 8 * hfl]);
        const uint4 q0 = ap[0], q1 = ap[1], q2 = ap[2], q3 = ap[3];
        auto at = [&](const unsigned lo, const unsigned hi2) { return xoff + (((unsigned long long)hi2 << 32) | lo); };
        const unsigned long long g0 = at(q0.x, q0.y), g1 = at(q0.z, q0.w), g2 = at(q1.x, q1.y), g3 = at(q1.z, q1.w);
        const unsigned long long g4 = at(q2.x, q2.y), g5 = at(q2.z, q2.w), g6 = at(q3.x, q3.y), g7 = at(q3.z, q3.w);
        const unsigned dst = ring_lds + 8192u * (unsigned)slot + 4096u;
        unsigned keep;
        asm volatile(
            "s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[l]\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %[g0], off offset:-4096\n\t"
            "global_load_lds_dwordx4 %[g1], off offset:-3072\n\t"
            "global_load_lds_dwordx4 %[g2], off offset:-2048\n\t"
            "global_load_lds_dwordx4 %[g3], off offset:-1024\n\t"
            "global_load_lds_dwordx4 %[g4], off\n\t"
            "global_load_lds_dwordx4 %[g5], off offset:1024\n\t"
            "global_load_lds_dwordx4 %[g6], off offset:2048\n\t"
            "global_load_lds_dwordx4 %[g7], off offset:3072\n\t"
            "s_mov_b32 m0, %[k]"
            : [k] "=&s"(keep)
            : [l] "s"(dst), [g0] "v"(g0), [g1] "v"(g1), [g2] "v"(g2), [g3] "v"(g3), [g4] "v"(g4), [g5] "v"(g5), [g6] "v"(g6), [g7] "v"(g7)
            : "memory");
      };
      auto consume = [&](const int st, const float cvr, float (&xs0)[16], float (&xs1)[16]) __attribute__((always_inline)) {
        const int ccnt = min(16, p2 - (p1 + 16 * st));
        const bool inl = (ln & 15) < ccnt;
        const float cvj = inl ? cvr : 0.f;
        csum += ln < 16 ? cvj : 0.f;
        const float f2 = inl ? sx : 0.f;                                                       // M2's operand scale
        float f1;                                                                              // M1's
        if constexpr (SYM) f1 = inl ? sx * __builtin_amdgcn_sqrtf(fmaxf(cvr - 1.f, 0.f) * sw_) : 0.f;
        else f1 = inl ? (cvr - 1.f) * sw_ : 0.f;   // (!SYM: the A side's extra factor on top of the B side's 2^e x)
        // the right-hand side from the raw vectors, which then become 2^e x IN PLACE (no second copy: three buffers of 32
        // registers are in flight or in use next to the 64 of M2's last four tiles)
#pragma unroll
        for (int s2 = 0; s2 < 16; s2++) {
          const float cv = readlane_f(cvj, s2);
          u0 = fmaf(cv, xs0[s2], u0);
          u1 = fmaf(cv, xs1[s2], u1);
          const float s_m = readlane_f(f2, s2);
          xs0[s2] *= s_m;
          xs1[s2] *= s_m;
        }
        asm volatile("" : "+v"(u0), "+v"(u1));
        __builtin_amdgcn_sched_barrier(0);
        float (&xm0)[16] = xs0;
        float (&xm1)[16] = xs1;
        // The operands stay in registers and the 40 matrix instructions follow the vector work.  SYM: one operand set for M1
        // (A side = B side).  Some confidence below 1: B side = 2^e x with both terms (M2 uses its leading term), A side =
        // 2^e' (c - 1) 2^e x.
        f16x8 ah[4], al[4], mh[4], bh[4], bl[4];
        auto m2_products = [&]() __attribute__((always_inline)) {   // (first: their operand dies before M1's are built)
          mf_sfor<4>([&](auto kt) {
            constexpr int K = decltype(kt)::value;
            mf_sfor<4 - K>([&](auto st2) {
              constexpr int I = K + decltype(st2)::value;
              constexpr int T = mf_tid(I, K);
              if constexpr (T < 6) mf_mma16<10 + T>(mh[K], mh[I]);
              else hi[T - 6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(mh[K], mh[I], hi[T - 6], 0, 0, 0);
            });
          });
        };
        // g = f1 / f2: the factor that turns M2's operand 2^e x into M1's (SYM: sqrt((c - 1) 2^e'); else (c - 1) 2^e')
        float gj;
        if constexpr (SYM) gj = inl ? __builtin_amdgcn_sqrtf(fmaxf(cvr - 1.f, 0.f) * sw_) : 0.f;
        else gj = f1;
        if constexpr (SYM) {
          cgm_operands_hi(xm0, mh[0], mh[1]);
          cgm_operands_hi(xm1, mh[2], mh[3]);
          __builtin_amdgcn_sched_barrier(0);
          m2_products();
          __builtin_amdgcn_sched_barrier(0);
        } else {
          mf_operands(xm0, bh[0], bl[0], bh[1], bl[1]);
          mf_operands(xm1, bh[2], bl[2], bh[3], bl[3]);
#pragma unroll
          for (int t = 0; t < 4; t++) mh[t] = bh[t];
          __builtin_amdgcn_sched_barrier(0);
          m2_products();
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s2 = 0; s2 < 16; s2++) {
          const float s_a = readlane_f(gj, s2);
          xs0[s2] *= s_a;
          xs1[s2] *= s_a;
        }
        mf_operands(xs0, ah[0], al[0], ah[1], al[1]);
        __builtin_amdgcn_sched_barrier(0);
        mf_operands(xs1, ah[2], al[2], ah[3], al[3]);
        __builtin_amdgcn_sched_barrier(0);
        mf_sfor<3>([&](auto pt) {
          constexpr int pr = decltype(pt)::value;
          mf_sfor<4>([&](auto kt) {
            constexpr int K = decltype(kt)::value;
            mf_sfor<4 - K>([&](auto st2) {
              constexpr int I = K + decltype(st2)::value;
              if constexpr (SYM) mf_mma16<mf_tid(I, K)>(pr == 2 ? al[K] : ah[K], pr == 1 ? al[I] : ah[I]);
              else mf_mma16<mf_tid(I, K)>(pr == 2 ? al[K] : ah[K], pr == 1 ? bl[I] : bh[I]);
            });
          });
        });
      };
      // ---- SYM: the pipelined step.  One wave per SIMD issues in order: with a step's vector work in front of its 40 matrix
      // instructions the two add up (measured: the wave spends ~4500 cycles per step, 1280 of them matrix-pipe time, and neither
      // more bytes in flight nor fewer stalls change that).  Here step s - 1's matrix instructions (operands P, registers) go
      // out one or two at a time between the pieces of step s's vector work, which builds N.  The vectors are read from the
      // ring directly in the operand order -- lane (n, hf) takes coordinate 32 I + n of the non-zeros 8 hf .. 8 hf + 7: no lane
      // swaps -- and the per-non-zero weights (c for the right-hand side, 2^e for M2's operand, sqrt((c - 1) 2^e') on top for
      // M1's) come from LDS in that order too, prepared once per chunk of 64: no v_readlane.
      CgmOps OA, OB;   // ping-pong: even steps read OA (their predecessor's operands) and build OB, odd steps the other way round
      if constexpr (SYM) {
        const f16x8 z = mf_pack(0u, 0u, 0u, 0u);
#pragma unroll
        for (int t = 0; t < 4; t++) { OA.ah[t] = z; OA.al[t] = z; OA.mh[t] = z; }   // (step 0 has no predecessor: its 40 instructions add zero)
      }
      auto mm = [&](auto it, const CgmOps& P) __attribute__((always_inline)) {
        constexpr CgmProd pd = cgm_prod(decltype(it)::value);
#if defined(CGM_ABL) && (CGM_ABL & 1)   // timing-only dev build: no matrix instructions
        return;
#endif
        if constexpr (pd.kind == 0) {
          if constexpr (pd.T < 6) cgm_mma16<10 + pd.T>(P.mh[pd.K], P.mh[pd.I]);
          else hi[pd.T - 6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(P.mh[pd.K], P.mh[pd.I], hi[pd.T - 6], 0, 0, 0);
        } else {
          cgm_mma16<pd.T>(pd.kind == 3 ? P.al[pd.K] : P.ah[pd.K], pd.kind == 2 ? P.al[pd.I] : P.ah[pd.I]);
        }
      };
      // chunk c has landed: its vectors' addresses (see request), M1's operand scale sqrt((c - 1) 2^e') next to the confidences,
      // their sum (the row's entries only)
      auto transform = [&](const int c) __attribute__((always_inline)) {
        const int cs = c % 3;
        int l2 = ln;   // (this sub-step's own copy: the LDS addresses below are recomputed from it, not carried through the loop in
        asm volatile("" : "+v"(l2));   //  registers -- hipcc spilled them, and a scratch reload is a counted load that drains the LDS-DMA queue)
        const float cvr = sw.mval[cs][l2];
        const int id = sw.midx[cs][l2];
        if constexpr (SYM) csum += p1 + 64 * c + l2 < p2 ? cvr : 0.f;
        wave_sync();
        sw.maddr[cs][l2] = (unsigned long long)(unsigned)id * (k * 4) - (long long)(((l2 & 7) - 4) * 1024);
        if constexpr (SYM) sw.gval[cs][l2] = __builtin_amdgcn_sqrtf(fmaxf(cvr - 1.f, 0.f) * sw_);
        wave_sync();
      };
      // the row's last step: the vectors beyond its end (copies of the last entry) become zeros in the ring -- no mask anywhere else
      auto zero_tail = [&](const int slot, const int ccnt) __attribute__((always_inline)) {
        float* rs = &sw.ring[slot][0];
#pragma unroll 1
        for (int jz = ccnt; jz < 16; jz++) {
          rs[(jz & 7) * 256 + (jz >> 3) * 128 + ln] = 0.f;
          rs[(jz & 7) * 256 + (jz >> 3) * 128 + 64 + ln] = 0.f;
        }
        wave_sync();
      };
      auto step_sym = [&](const int slot, const int cs, auto jt, const CgmOps& P, CgmOps& N, const float4 va0, const float4 vb0, const float2 wc0, const float2 wg0) __attribute__((always_inline)) {
        constexpr int j = decltype(jt)::value;
        const float* rs = &sw.ring[slot][0] + 128 * hfl + 4 * (ln & 31);
        // (the weights of a pair of non-zeros are read with the pair's vectors: sixteen of them held through the step were the
        // registers hipcc spilled -- a scratch reload is a counted load, it drained the LDS-DMA queue)
        const float2* pc = reinterpret_cast<const float2*>(&sw.mval[cs][16 * j + 8 * hfl]);
        const float2* pg = reinterpret_cast<const float2*>(&sw.gval[cs][16 * j + 8 * hfl]);
        float2 wc[2], wg[2];
        wc[0] = wc0;
        wg[0] = wg0;
        // unit (q, I): the non-zeros 2 q and 2 q + 1 (of this half's eight) at coordinate block I; a pair's two reads -- the lane's
        // four blocks of one vector each -- go out while its predecessor is worked on
        float4 va[2], vb[2];
        va[0] = va0;
        vb[0] = vb0;
#if defined(CGM_ABL) && (CGM_ABL & 2)   // timing-only dev build: the matrix instructions alone
        mf_sfor<40>([&](auto it) { mm(it, P); __builtin_amdgcn_sched_barrier(0); });
        N = P;
        return;
#endif
        unsigned hm[4][4], hh[4][4], ll[4][4];   // [block][pair]
        mf_sfor<4>([&](auto qt) {
          constexpr int q = decltype(qt)::value;
          if constexpr (q < 3) {
            va[(q + 1) & 1] = *reinterpret_cast<const float4*>(rs + (2 * q + 2) * 256);
            vb[(q + 1) & 1] = *reinterpret_cast<const float4*>(rs + (2 * q + 3) * 256);
            wc[(q + 1) & 1] = pc[q + 1];
            wg[(q + 1) & 1] = pg[q + 1];
          }
          const float c0 = wc[q & 1].x, c1 = wc[q & 1].y, g0 = wg[q & 1].x, g1 = wg[q & 1].y;
          const float4 xa = va[q & 1], xb = vb[q & 1];
          const float a0s[4] = {xa.x, xa.y, xa.z, xa.w}, a1s[4] = {xb.x, xb.y, xb.z, xb.w};
          mf_sfor<4>([&](auto It) {
            constexpr int I = decltype(It)::value, u = 4 * q + I, m0 = (5 * u) / 2, m1 = (5 * (u + 1)) / 2;
            const float a0 = a0s[I], a1 = a1s[I];
            ua[I] = fmaf(c1, a1, fmaf(c0, a0, ua[I]));
            const float t0 = a0 * sx, t1 = a1 * sx;
            {
              const f32x2 tv = {t0, t1};
              hm[I][q] = __builtin_bit_cast(unsigned, __builtin_convertvector(tv, f16x2));
            }
            mm(std::integral_constant<int, m0>{}, P);
            __builtin_amdgcn_sched_barrier(0);
            {
              // M1's two fp16 terms of t g: the leading one rounds the product, the second takes the EXACT remainder
              // fma(t, g, -hi) (v_fma_mix_f32: the fp16 half is an operand; one instruction where a conversion and a subtraction were two)
              const f32x2 yv = {t0 * g0, t1 * g1};
              const unsigned hv = __builtin_bit_cast(unsigned, __builtin_convertvector(yv, f16x2));
              float r0, r1;
              asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(t0), "v"(g0), "v"(hv));
              asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(t1), "v"(g1), "v"(hv));
              const f32x2 rv = {r0, r1};
              hh[I][q] = hv;
              ll[I][q] = __builtin_bit_cast(unsigned, __builtin_convertvector(rv, f16x2));
            }
            mm(std::integral_constant<int, m0 + 1>{}, P);
            if constexpr (m1 - m0 == 3) {
              __builtin_amdgcn_sched_barrier(0);
              mm(std::integral_constant<int, m0 + 2>{}, P);
            }
            __builtin_amdgcn_sched_barrier(0);
          });
        });
#pragma unroll
        for (int I = 0; I < 4; I++) {
          N.mh[I] = mf_pack(hm[I][0], hm[I][1], hm[I][2], hm[I][3]);
          N.ah[I] = mf_pack(hh[I][0], hh[I][1], hh[I][2], hh[I][3]);
          N.al[I] = mf_pack(ll[I][0], ll[I][1], ll[I][2], ll[I][3]);
        }
      };
      // prologue: chunks 0 and 1 of the indices, then steps 0 and 1
      CGM_TICK(0)
      CGM_COUNT(8, nsteps) CGM_COUNT(9, 1)
      wave_sync();
      request_meta(0);
      request_meta(1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      transform(0);
      request(0, 0);
      request(1, 1);
      const int tail_st = ((p2 - p1) & 15) ? nsteps - 1 : -1;
      CGM_TICK(1)
      int slot = 0;   // = st % 3
      for (int c = 0; 4 * c < nsteps; c++) {
        bool done = false;
        mf_sfor<4>([&](auto jt) {
          constexpr int j = decltype(jt)::value;
          const int st = 4 * c + j;
          if (done || st >= nsteps) { done = true; return; }   // wave-uniform
          const int slot2 = slot == 0 ? 2 : slot - 1;           // (st + 2) % 3
          float4 fva, fvb;   // the step's first pair of vectors and its weights
          float2 fwc, fwg;
          auto first_reads = [&](const int sl, const int cs, const int jj) __attribute__((always_inline)) {
            const float* rs = &sw.ring[sl][0] + 128 * hfl + 4 * (ln & 31);
            fva = *reinterpret_cast<const float4*>(rs);
            fvb = *reinterpret_cast<const float4*>(rs + 256);
            fwc = *reinterpret_cast<const float2*>(&sw.mval[cs][16 * jj + 8 * hfl]);
            fwg = *reinterpret_cast<const float2*>(&sw.gval[cs][16 * jj + 8 * hfl]);
          };
          if constexpr (j == 1) transform(c + 1);   // (landed two sub-steps ago at the latest; its first request is sub-step j = 2's)
          const bool more = st + 2 < nsteps;
          if (more) {
            // younger than step st in the queue: step st + 1 (8) and, behind sub-step j = 0, a chunk of indices (2)
            if constexpr (j == 1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            CGM_TICK(2)
            if constexpr (SYM) first_reads(slot, c % 3, j);   // (their latency passes while the requests below are issued)
            // (measured and not kept: the eight pieces one by one BETWEEN the step's matrix instructions, like the vector work:
            // 16.8 ms per launch against 15.0 -- a piece costs more there than the ~55 cycles it takes in a block in front)
            request(st + 2, slot2);
            if constexpr (j == 0) request_meta(c + 2);
            CGM_TICK(3)
          } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            CGM_TICK(2)
            if constexpr (SYM) {
              if (st == tail_st) zero_tail(slot, (p2 - p1) & 15);   // wave-uniform (the row's last step is never in the branch above)
              first_reads(slot, c % 3, j);
            }
          }
          if constexpr (SYM) {
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (j & 1) step_sym(slot, c % 3, jt, OB, OA, fva, fvb, fwc, fwg);
            else step_sym(slot, c % 3, jt, OA, OB, fva, fvb, fwc, fwg);
          } else {
            const float cvr = sw.mval[c % 3][16 * j + (ln & 15)];
            const float* rs = &sw.ring[slot][0];
            float xs0[16], xs1[16];
#pragma unroll
            for (int s2 = 0; s2 < 16; s2++) {
              xs0[s2] = rs[(s2 & 7) * 256 + (s2 >> 3) * 128 + 4 * (ln & 31) + hfl];       // (block hfl, row ln & 31)
              xs1[s2] = rs[(s2 & 7) * 256 + (s2 >> 3) * 128 + 4 * (ln & 31) + 2 + hfl];   // (block 2 + hfl)
            }
            __builtin_amdgcn_sched_barrier(0);
            consume(st, cvr, xs0, xs1);
            __builtin_amdgcn_sched_barrier(0);
          }
          slot = slot == 2 ? 0 : slot + 1;
          CGM_TICK(4)
        });
        if (done) break;
      }
      if constexpr (SYM) {   // the last step's matrix instructions; the right-hand side: the halves' partials added
        if (nsteps & 1) mf_sfor<40>([&](auto it) { mm(it, OB); });   // (wave-uniform: which set the last step built)
        else mf_sfor<40>([&](auto it) { mm(it, OA); });
#pragma unroll
        for (int I = 0; I < 4; I++) {
          const unsigned du = __float_as_uint(ua[I]);
          const auto sw2 = __builtin_amdgcn_permlane32_swap(du, du, false, false);
          ua[I] = __uint_as_float(sw2[0]) + __uint_as_float(sw2[1]);
        }
      }
      wave_sync();   // (the ring's LDS becomes the solve's)
    }
    // ---- unscale (powers of two: exact); the right-hand side and the warm start per lane = row ----
    ln = lane;
    asm volatile("" : "+v"(ln));
    const int n = ln & 31, hf = ln >> 5;
    const float un_m1 = (un1 * un1) * (SYM ? unw : unw), un_m2 = un1 * un1;
    MF_DRAIN();
    mf_sfor<160>([&](auto rt) {
      constexpr int R = decltype(rt)::value;
      mf_wr<R>(mf_rd<R>() * un_m1);
    });
    mf_sfor<96>([&](auto rt) {
      constexpr int R = 160 + decltype(rt)::value;
      mf_wr<R>(mf_rd<R>() * un_m2);
    });
    {
      float um = un_m2;   // (a register of this point's own: held as a pair from the kernel's first lines it was the one spill)
      asm volatile("" : "+v"(um));
#pragma unroll
      for (int t = 0; t < 4; t++)
#pragma unroll
        for (int e = 0; e < 16; e++) hi[t][e] *= um;
    }
    float b[4], x[4];
    if constexpr (SYM) {
#pragma unroll
      for (int I = 0; I < 4; I++) {
        b[I] = ua[I];
        x[I] = sw.x0[4 * n + I];
      }
    } else {
      wave_sync();
      sw.rhs[ln] = u0;
      sw.rhs[64 + ln] = u1;
      wave_sync();
#pragma unroll
      for (int I = 0; I < 4; I++) {
        b[I] = sw.rhs[32 * I + n];
        x[I] = sw.x0[4 * n + I];
      }
    }
    auto rd_m1 = [&](auto tt, auto vt) { return mf_rd<16 * decltype(tt)::value + decltype(vt)::value>(); };
    auto g_tile = [&](auto tt, float (&g)[16]) {
      constexpr int T = decltype(tt)::value;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const float4 t = sm.G[T][ln][q];
        g[4 * q] = t.x; g[4 * q + 1] = t.y; g[4 * q + 2] = t.z; g[4 * q + 3] = t.w;
      }
    };
    // ---- cg_solver_implicit (wrmf_implicit.hpp:8-32) on A = XtX + M1 ----
    CGM_TICK(5)
    float r[4], p[4], ap[4];
    cgm_matvec(rd_m1, g_tile, sw, n, hf, ln, x, ap);
#pragma unroll
    for (int I = 0; I < 4; I++) { r[I] = b[I] - ap[I]; p[I] = r[I]; }
    double rsold = (double)cgm_dot(r, r, hf);
    for (int stp = 0; stp < a.cg_steps; stp++) {
      cgm_matvec(rd_m1, g_tile, sw, n, hf, ln, p, ap);
      const float alpha = (float)(rsold / (double)cgm_dot(p, ap, hf));
#pragma unroll
      for (int I = 0; I < 4; I++) { x[I] = fmaf(alpha, p[I], x[I]); r[I] = fmaf(-alpha, ap[I], r[I]); }
      const double rsnew = (double)cgm_dot(r, r, hf);
      if (rsnew < (double)kCgTolMf) break;
      const float beta = (float)(rsnew / rsold);
#pragma unroll
      for (int I = 0; I < 4; I++) p[I] = fmaf(p[I], beta, r[I]);
      rsold = rsnew;
    }
    // ---- the row and its loss term: sum c - 2 y.b + y^T (M1 + M2) y + lambda |y|^2 ----
    CGM_TICK(6)
    if (hf == 0) {
      int row2 = row;   // (the row's address again from scalar registers: held in a vector pair since the warm start it was the one spill)
      asm volatile("" : "+s"(row2));
      float* yr = a.Y + (size_t)row2 * k;
      *reinterpret_cast<float4*>(yr + 4 * n) = make_float4(x[0], x[1], x[2], x[3]);
    }
    auto rd_m2 = [&](auto tt, auto vt) {
      constexpr int T = decltype(tt)::value, v = decltype(vt)::value;
      if constexpr (T < 6) return mf_rd<160 + 16 * T + v>();
      else return hi[T - 6][v];
    };
    // y^T (M1 + M2) y from the lower tiles: sum over the tiles of y_I(n) * (tile row . y_K), off-diagonal tiles twice -- the
    // "direct" half of a product only, both matrices in one pass
    float yMy;
    {
      wave_sync();
      if (hf == 0) {
#pragma unroll
        for (int I = 0; I < 4; I++) sw.vec[32 * I + n] = x[I];
      }
      wave_sync();
      float part = 0.f;
      mf_sfor<4>([&](auto kt) {
        constexpr int K = decltype(kt)::value;
        float pc[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const float4 t = *reinterpret_cast<const float4*>(&sw.vec[32 * K + 8 * q + 4 * hf]);
          pc[4 * q] = t.x; pc[4 * q + 1] = t.y; pc[4 * q + 2] = t.z; pc[4 * q + 3] = t.w;
        }
        mf_sfor<4 - K>([&](auto st) {
          constexpr int I = K + decltype(st)::value;
          constexpr int T = mf_tid(I, K);
          float d = 0.f;
          mf_sfor<16>([&](auto vt) {
            constexpr int v = decltype(vt)::value;
            d = fmaf(rd_m1(std::integral_constant<int, T>{}, vt) + rd_m2(std::integral_constant<int, T>{}, vt), pc[v], d);
          });
          part = fmaf(I == K ? d : 2.f * d, x[I], part);
        });
      });
      yMy = wave_sum(part);   // (both halves of the wave hold half of every row's columns: the sum over all 64 lanes is the form)
    }
    const float yb = cgm_dot(x, b, hf), yy = cgm_dot(x, x, hf);
    const float sc = wave_sum(csum);
    if (lane == 0) sw.loss += ((double)sc - 2.0 * (double)yb + (double)yMy) + a.lambda_loss * (double)yy;
    wave_sync();
  }
  if (lane == 0) a.loss_partials[loss_slot0 + 4 * blockIdx.x + wv] = sw.loss;
#ifdef RSP_MF_PROF
  CGM_TICK(7)
  pt[10] = __builtin_amdgcn_s_memtime() - pt_start;
  pt[11] = __builtin_amdgcn_s_memrealtime() - rt_start;
  if (a.ne_prof && lane == 0)
    for (int j = 0; j < 12; j++) atomicAdd(a.ne_prof + 16 + j, pt[j]);
#endif
}

}  // namespace
}  // namespace rsparse_hip

#define CGM_KERNEL(NAME, SYM)                                                                                                     \
  extern "C" __global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(128))) void NAME(                                  \
      rsparse_hip::AlsArgs a, const int32_t* __restrict__ rows, int n_rows, int loss_slot0) {                                       \
    rsparse_hip::als_cg_mf_body<SYM>(a, rows, n_rows, loss_slot0);                                                                  \
  }
CGM_KERNEL(rsparse_hip_als_cg_mf_implicit, true)
CGM_KERNEL(rsparse_hip_als_cg_mf_implicit_any, false)
#undef CGM_KERNEL

namespace rsparse_hip {

bool cg_mf_supported(int k, bool implicit) { return implicit && k == 128; }
int cg_mf_grid(int n_rows) { return std::max(1, std::min((n_rows + 3) / 4, kCgMfGrid)); }
// two launches (see the kernel's first lines), each with one slot per wave
int cg_mf_loss_slots(int n_rows) { return 2 * 4 * cg_mf_grid(n_rows); }

hipError_t launch_als_cg_mf(const AlsArgs& a, const int32_t* rows, int n_rows, int loss_slot0, hipStream_t s, hipEvent_t* ev_slot) {
  if (n_rows <= 0) return hipSuccess;
  if (!a.ne_stats || !cg_mf_supported(a.k, true) || (reinterpret_cast<uintptr_t>(a.X) & 3)) return hipErrorInvalidValue;
  const int grid = cg_mf_grid(n_rows);
  auto kern = rsparse_hip_als_cg_mf_implicit;
  auto kern2 = rsparse_hip_als_cg_mf_implicit_any;
  const int b1 = (int)sizeof(CgmSmem<12>), b2 = (int)sizeof(CgmSmem<20>);
  hipError_t e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, b1)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern2), hipFuncAttributeMaxDynamicSharedMemorySize, b2)) != hipSuccess) return e;
  prof_note(ev_slot, reinterpret_cast<const void*>(kern));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), b1, s, a, rows, n_rows, loss_slot0);
  hipLaunchKernelGGL(kern2, dim3(grid), dim3(256), b2, s, a, rows, n_rows, loss_slot0 + 4 * grid);
  return hipGetLastError();
}

}  // namespace rsparse_hip
