// WRMF / ALS kernels for MI355X (gfx950, wave64).  Hand-written HIP; no CUDA shims, no dual paths.
//
// What each kernel replaces in the reference (paths relative to /root/reference):
//   als_cg_short_kernel / als_cg_long_kernel
//        the OpenMP column loop of als_implicit<T> / als_explicit<T> with solver == CG
//        (inst/include/wrmf_implicit.hpp:162-197,254-283 + cg_solver_implicit :8-32;
//         inst/include/wrmf_explicit.hpp:68-100,128-144 + cg_solver_explicit :8-31)
//   gramian_partial_kernel / gramian_reduce_kernel
//        XtX = tcrossprod(X) + fl(lambda) I done on the R side with threaded BLAS
//        (R/model_WRMF.R:474-486, :347-353) -- the one MFMA kernel
//   weighted_sumsq_kernel
//        lambda * accu(X % X) / accu((X % X) * cnt_X)  (wrmf_implicit.hpp:299-301,
//        wrmf_explicit.hpp:160-170)
//
// Scheduling: a row (= CSC column) with at most T = 32 non-zeros is solved by ONE wavefront with its
// gathered factor vectors resident in a per-wave LDS tile for the whole CG solve (short kernel;
// rows are claimed dynamically from a per-workgroup LDS counter, the analogue of the reference's
// `schedule(dynamic)`).  Longer rows are solved by one WORKGROUP per row (long kernel): its 4 waves
// own interleaved 32-nnz chunks (resident when the row has <= 128 non-zeros, re-streamed through L2
// otherwise), split the k x k Gramian product by k-range, and combine partial vectors through LDS
// with one barrier per CG sweep.  Rows are taken longest-first.
#include "wrmf_internal.h"
#include "wrmf_device.h"

namespace rsparse_hip {
namespace {

using namespace dev;

constexpr float kCgTol = 1e-10f;  // CG_TOL, inst/include/wrmf.hpp:22

template <int KP, int T, int W, bool IMPLICIT>
struct CgSmem {
  static constexpr int LDT = Geo<KP>::LDT;
  static constexpr size_t gram_floats = IMPLICIT ? (size_t)KP * KP : 0;
  static constexpr size_t tile_floats = (size_t)W * T * LDT;
  static constexpr size_t vec_floats = (size_t)W * KP;
  static constexpr size_t red_floats = (size_t)2 * W * KP + 2 * W;  // long kernel only
  static constexpr size_t short_bytes = (gram_floats + tile_floats + vec_floats) * 4 + 16;
  static constexpr size_t long_bytes = (gram_floats + tile_floats + vec_floats + red_floats) * 4 + 16;
};

template <int KP, int W>
__device__ __forceinline__ void load_gram_lds(float* sG, const float* __restrict__ G, int k, int tid) {
  // sG[r][c] = G[r*k + c] (symmetric), zero padded to KP x KP
  for (int e = tid; e < KP * KP; e += W * 64) {
    const int r = e / KP, c = e % KP;
    sG[e] = (r < k && c < k) ? G[(size_t)r * k + c] : 0.f;
  }
}

// ------------------------------------------------------------------------------------------------
// short rows: one wavefront per row
// ------------------------------------------------------------------------------------------------
template <int KP, int T, int W, bool IMPLICIT, bool VEC>
__global__ __launch_bounds__(W * 64) void als_cg_short_kernel(AlsArgs a) {
  constexpr int EPL = Geo<KP>::EPL, LDT = Geo<KP>::LDT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sG = reinterpret_cast<float*>(smem);
  float* sTiles = sG + CgSmem<KP, T, W, IMPLICIT>::gram_floats;
  float* sVecs = sTiles + CgSmem<KP, T, W, IMPLICIT>::tile_floats;
  int* sCtr = reinterpret_cast<int*>(sVecs + CgSmem<KP, T, W, IMPLICIT>::vec_floats);

  const int tid = threadIdx.x, lane = tid & 63, wv = rfl(tid >> 6);
  const int k = a.k;
  if constexpr (IMPLICIT) load_gram_lds<KP, W>(sG, a.XtX, k, tid);
  for (int e = tid; e < (int)(CgSmem<KP, T, W, IMPLICIT>::tile_floats + CgSmem<KP, T, W, IMPLICIT>::vec_floats);
       e += W * 64)
    sTiles[e] = 0.f;
  if (tid == 0) *sCtr = 0;
  __syncthreads();

  float* tile = sTiles + wv * T * LDT;
  float* vec = sVecs + wv * KP;
  const int row0 = blockIdx.x * kRowsPerWGShort;
  const int row_end = min(a.n_cols, row0 + kRowsPerWGShort);
  const int e0 = lane * EPL;
  const bool active = e0 < KP;
  const int jl = lane % T;
  double wloss = 0.0;
  // implicit feedback with a global bias (cg_solver_implicit_global_bias, wrmf_implicit.hpp:35-57,203): a.gbias != 0, then
  // the first residual takes  + global_bias inside c1 % (.)  and  + a.rhs_init (global_bias_base), empty rows are solved
  // (:178) and the loss compares with a.loss_tgt_const = 1 - global_bias (:262-264)
  const bool gb = IMPLICIT && a.gbias != 0.f;
  const float gbias = gb ? a.gbias : 0.f, ltgt = gb ? a.loss_tgt_const : 1.f;

  while (true) {
    int rr = 0;
    if (lane == 0) rr = atomicAdd(sCtr, 1);
    const int row = row0 + rfl(rr);
    if (row >= row_end) break;
    const int p1 = rfl(a.col_ptrs[row]), p2 = rfl(a.col_ptrs[row + 1]);
    const int cnt = p2 - p1;
    if (cnt > T) continue;  // solved by the long-row kernel
    float* yrow = a.Y + (size_t)row * k;
    if (cnt <= 0 && !gb) {  // empty column -> zeros (wrmf_implicit.hpp:281, wrmf_explicit.hpp:142)
      for (int e = lane; e < k; e += 64) yrow[e] = 0.f;
      continue;
    }
    const bool jvalid = jl < cnt;
    const int myidx = jvalid ? a.row_idx[p1 + jl] : 0;
    const float c = jvalid ? a.vals[p1 + jl] : 0.f;
    wave_sync();
    gather_chunk<KP, T, VEC>(a.X, k, myidx, cnt, tile, lane);

    float x[EPL], r[EPL], p[EPL], ap[EPL];
#pragma unroll
    for (int u = 0; u < EPL; u++) x[u] = (active && e0 + u < k) ? yrow[e0 + u] : 0.f;  // warm start
    const float lam_use =
        IMPLICIT ? 0.f : (float)(a.lambda_loss * (a.dynamic_lambda ? (double)(float)cnt : 1.0));

    // r = X_nnz (c - c1 % (X_nnz^T x)) - XtX x        | explicit: X_nnz (c - X_nnz^T x) - lambda x
    put_vec<KP>(vec, x, lane, active);
    wave_sync();
    float t = tile_dot<KP, T>(tile, vec, lane);
    float wgt = jvalid ? (IMPLICIT ? c - (c - 1.f) * (t + gbias) : c - t) : 0.f;
#pragma unroll
    for (int u = 0; u < EPL; u++) r[u] = 0.f;
    tile_axpy<KP>(tile, wgt, cnt, lane, active, r);
    if constexpr (IMPLICIT) {
      gram_mv<KP>(sG, vec, 0, KP, lane, active, -1.f, r);
      if (gb) {
#pragma unroll
        for (int u = 0; u < EPL; u++) r[u] += (active && e0 + u < k) ? a.rhs_init[e0 + u] : 0.f;
      }
    } else {
#pragma unroll
      for (int u = 0; u < EPL; u++) r[u] -= lam_use * x[u];
    }
    float rs = 0.f;
#pragma unroll
    for (int u = 0; u < EPL; u++) { p[u] = r[u]; rs = fmaf(r[u], r[u], rs); }
    float rsold = wave_sum(rs);

    for (int it = 0; it < a.cg_steps; ++it) {
      wave_sync();
      put_vec<KP>(vec, p, lane, active);
      wave_sync();
      t = tile_dot<KP, T>(tile, vec, lane);
      wgt = jvalid ? (IMPLICIT ? (c - 1.f) * t : t) : 0.f;
#pragma unroll
      for (int u = 0; u < EPL; u++) ap[u] = 0.f;
      tile_axpy<KP>(tile, wgt, cnt, lane, active, ap);
      if constexpr (IMPLICIT) {
        gram_mv<KP>(sG, vec, 0, KP, lane, active, 1.f, ap);
      } else {
#pragma unroll
        for (int u = 0; u < EPL; u++) ap[u] = fmaf(lam_use, p[u], ap[u]);
      }
      float pap = 0.f;
#pragma unroll
      for (int u = 0; u < EPL; u++) pap = fmaf(p[u], ap[u], pap);
      pap = wave_sum(pap);
      const float alpha = (float)((double)rsold / (double)pap);   // double scalars like the reference (wrmf_implicit.hpp:18)
      rs = 0.f;
#pragma unroll
      for (int u = 0; u < EPL; u++) {
        x[u] = fmaf(alpha, p[u], x[u]);
        r[u] = fmaf(-alpha, ap[u], r[u]);
        rs = fmaf(r[u], r[u], rs);
      }
      const float rsnew = wave_sum(rs);
      if (rsnew < kCgTol) break;
      const float beta = (float)((double)rsnew / (double)rsold);
#pragma unroll
      for (int u = 0; u < EPL; u++) p[u] = fmaf(p[u], beta, r[u]);
      rsold = rsnew;
    }

    // loss row term and write-back
    wave_sync();
    put_vec<KP>(vec, x, lane, active);
    wave_sync();
    t = tile_dot<KP, T>(tile, vec, lane);
    const float d = IMPLICIT ? ltgt - t : c - t;
    const float lj = (jvalid && lane < T) ? (IMPLICIT ? c * d * d : d * d) : 0.f;
    const float rl = wave_sum(lj);
    float xx = 0.f;
#pragma unroll
    for (int u = 0; u < EPL; u++) xx = fmaf(x[u], x[u], xx);
    xx = wave_sum(xx);
    wloss += IMPLICIT ? (double)rl + a.lambda_loss * (double)xx : (double)(rl + lam_use * xx);
#pragma unroll
    for (int u = 0; u < EPL; u++)
      if (active && e0 + u < k) yrow[e0 + u] = x[u];
  }
  if (lane == 0) a.loss_partials[(size_t)blockIdx.x * W + wv] = wloss;
}

// ------------------------------------------------------------------------------------------------
// long rows: one workgroup per row
// ------------------------------------------------------------------------------------------------
template <int KP, int T, int W, bool IMPLICIT, bool VEC>
__global__ __launch_bounds__(W * 64) void als_cg_long_kernel(AlsArgs a, size_t loss_slot0) {
  constexpr int EPL = Geo<KP>::EPL, LDT = Geo<KP>::LDT;
  static_assert(KP % (4 * W) == 0, "Gramian k-range must split evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sG = reinterpret_cast<float*>(smem);
  float* sTiles = sG + CgSmem<KP, T, W, IMPLICIT>::gram_floats;
  float* sVecs = sTiles + CgSmem<KP, T, W, IMPLICIT>::tile_floats;
  float* sRed = sVecs + CgSmem<KP, T, W, IMPLICIT>::vec_floats;  // [2][W][KP]
  float* sRedL = sRed + 2 * W * KP;                               // [2][W]

  const int tid = threadIdx.x, lane = tid & 63, wv = rfl(tid >> 6);
  const int k = a.k;
  if constexpr (IMPLICIT) load_gram_lds<KP, W>(sG, a.XtX, k, tid);
  for (int e = tid; e < (int)(CgSmem<KP, T, W, IMPLICIT>::tile_floats + CgSmem<KP, T, W, IMPLICIT>::vec_floats +
                              CgSmem<KP, T, W, IMPLICIT>::red_floats);
       e += W * 64)
    sTiles[e] = 0.f;
  __syncthreads();

  float* tile = sTiles + wv * T * LDT;
  float* vec = sVecs + wv * KP;
  const int e0 = lane * EPL;
  const bool active = e0 < KP;
  const int jl = lane % T;
  int buf = 0;
  double wloss = 0.0;
  const bool gb = IMPLICIT && a.gbias != 0.f;   // global bias, see als_cg_short_kernel
  const float gbias = gb ? a.gbias : 0.f, ltgt = gb ? a.loss_tgt_const : 1.f;

  for (int li = blockIdx.x; li < a.n_long; li += gridDim.x) {
    const int row = rfl(a.long_rows[li]);
    const int p1 = rfl(a.col_ptrs[row]), p2 = rfl(a.col_ptrs[row + 1]);
    const int cnt = p2 - p1;
    const int nchunks = (cnt + T - 1) / T;
    const bool resident = nchunks <= W;
    float* yrow = a.Y + (size_t)row * k;
    const float lam_use =
        IMPLICIT ? 0.f : (float)(a.lambda_loss * (a.dynamic_lambda ? (double)(float)cnt : 1.0));

    // chunk state (valid for the whole solve when resident)
    int ccnt = 0, cidx = 0;
    float cval = 0.f;
    if (resident && wv < nchunks) {
      const int base = p1 + wv * T;
      ccnt = min(T, p2 - base);
      cidx = jl < ccnt ? a.row_idx[base + jl] : 0;
      cval = jl < ccnt ? a.vals[base + jl] : 0.f;
      wave_sync();
      gather_chunk<KP, T, VEC>(a.X, k, cidx, ccnt, tile, lane);
      wave_sync();
    }

    float x[EPL], r[EPL], p[EPL], ap[EPL];
#pragma unroll
    for (int u = 0; u < EPL; u++) x[u] = (active && e0 + u < k) ? yrow[e0 + u] : 0.f;

    // mode 0: out = X_nnz (c - c1 % X_nnz^T v) - G v ; mode 1: out = X_nnz (c1 % X_nnz^T v) + G v
    // mode 2: loss = sum_j c_j (1 - t_j)^2  (explicit: weights 1 and residual c_j - t_j)
    auto sweep = [&](const float(&v)[EPL], const int mode, float(&out)[EPL], float& loss_out) {
      wave_sync();
      put_vec<KP>(vec, v, lane, active);
      wave_sync();
      float acc[EPL];
#pragma unroll
      for (int u = 0; u < EPL; u++) acc[u] = 0.f;
      float lacc = 0.f;
      for (int ch = wv; ch < nchunks; ch += W) {
        if (!resident) {
          const int base = p1 + ch * T;
          ccnt = min(T, p2 - base);
          cidx = jl < ccnt ? a.row_idx[base + jl] : 0;
          cval = jl < ccnt ? a.vals[base + jl] : 0.f;
          wave_sync();
          gather_chunk<KP, T, VEC>(a.X, k, cidx, ccnt, tile, lane);
          wave_sync();
        }
        const bool jvalid = jl < ccnt;
        const float t = tile_dot<KP, T>(tile, vec, lane);
        if (mode == 2) {
          const float d = IMPLICIT ? ltgt - t : cval - t;
          lacc += (jvalid && lane < T) ? (IMPLICIT ? cval * d * d : d * d) : 0.f;
        } else {
          float wgt;
          if (mode == 0) wgt = IMPLICIT ? cval - (cval - 1.f) * (t + gbias) : cval - t;
          else wgt = IMPLICIT ? (cval - 1.f) * t : t;
          wgt = jvalid ? wgt : 0.f;
          tile_axpy<KP>(tile, wgt, ccnt, lane, active, acc);
        }
      }
      float* red = sRed + buf * W * KP;
      if (mode == 2) {
        lacc = wave_sum(lacc);
        if (lane == 0) sRedL[buf * W + wv] = lacc;
      } else {
        if constexpr (IMPLICIT)
          gram_mv<KP>(sG, vec, wv * (KP / W), (wv + 1) * (KP / W), lane, active, mode == 0 ? -1.f : 1.f, acc);
        put_vec<KP>(red + wv * KP, acc, lane, active);
      }
      __syncthreads();
      if (mode == 2) {
        float s = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < W; w2++) s += sRedL[buf * W + w2];
        loss_out = s;
      } else {
#pragma unroll
        for (int u = 0; u < EPL; u++) out[u] = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < W; w2++) {
          if constexpr (EPL == 2) {
            const float2 q = *reinterpret_cast<const float2*>(red + w2 * KP + lane * 2);
            out[0] += q.x;
            out[1] += q.y;
          } else {
            out[0] += active ? red[w2 * KP + lane] : 0.f;
          }
        }
        if constexpr (!IMPLICIT) {
#pragma unroll
          for (int u = 0; u < EPL; u++) out[u] = fmaf(mode == 0 ? -lam_use : lam_use, v[u], out[u]);
        } else if (gb && mode == 0) {
#pragma unroll
          for (int u = 0; u < EPL; u++) out[u] += (active && e0 + u < k) ? a.rhs_init[e0 + u] : 0.f;
        }
      }
      buf ^= 1;
    };

    float dummy = 0.f;
    sweep(x, 0, r, dummy);
    float rs = 0.f;
#pragma unroll
    for (int u = 0; u < EPL; u++) { p[u] = r[u]; rs = fmaf(r[u], r[u], rs); }
    float rsold = wave_sum(rs);
    for (int it = 0; it < a.cg_steps; ++it) {
      sweep(p, 1, ap, dummy);
      float pap = 0.f;
#pragma unroll
      for (int u = 0; u < EPL; u++) pap = fmaf(p[u], ap[u], pap);
      pap = wave_sum(pap);
      const float alpha = (float)((double)rsold / (double)pap);   // double scalars like the reference (wrmf_implicit.hpp:18)
      rs = 0.f;
#pragma unroll
      for (int u = 0; u < EPL; u++) {
        x[u] = fmaf(alpha, p[u], x[u]);
        r[u] = fmaf(-alpha, ap[u], r[u]);
        rs = fmaf(r[u], r[u], rs);
      }
      const float rsnew = wave_sum(rs);
      if (rsnew < kCgTol) break;  // identical in every wave of the workgroup
      const float beta = (float)((double)rsnew / (double)rsold);
#pragma unroll
      for (int u = 0; u < EPL; u++) p[u] = fmaf(p[u], beta, r[u]);
      rsold = rsnew;
    }
    float rl = 0.f;
    sweep(x, 2, ap, rl);
    float xx = 0.f;
#pragma unroll
    for (int u = 0; u < EPL; u++) xx = fmaf(x[u], x[u], xx);
    xx = wave_sum(xx);
    if (wv == 0) {
      wloss += IMPLICIT ? (double)rl + a.lambda_loss * (double)xx : (double)(rl + lam_use * xx);
#pragma unroll
      for (int u = 0; u < EPL; u++)
        if (active && e0 + u < k) yrow[e0 + u] = x[u];
    }
  }
  if (tid == 0) a.loss_partials[loss_slot0 + blockIdx.x] = wloss;
}

// ------------------------------------------------------------------------------------------------
// Gramian  G = X X^T (+ ridge I),  X is k x n column-major: fp32 MFMA 32x32x2, lower-triangular
// 32x32 tiles only, one partial per wave, deterministic two-stage reduction.
// ------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KP>
__global__ __launch_bounds__(256, 2) void gramian_partial_kernel(const float* __restrict__ X, int k, int64_t n,
                                                              float* __restrict__ partials,
                                                              unsigned* __restrict__ absmax_bits) {
  constexpr int NT = KP / 32, U = 4;
  const int lane = threadIdx.x & 63;
  const int64_t gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
  const int col = lane & 31, half = lane >> 5;
  f32x16 acc[NT][NT];
#pragma unroll
  for (int i = 0; i < NT; i++)
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  float amax = 0.f;   // max |x| of what this lane reads: the matrix is read here anyway (wrmf_ne.hip wants its scale)
  // Software pipeline (round 5): the next trip's 16 loads are in flight while this trip's 40 matrix instructions run -- one trip
  // was load, wait, multiply: 2 us of memory latency in front of 1.1 us of MFMA, 2.7 ms for the 10 M x 128 Gramian of config 3.
  // The loads are unconditional (clamped address, 0 / 1 mask multiplied in: a select's load is sunk back under its condition).
  auto fetch = [&](const int64_t n0, float (&dst)[U][NT]) {   // raw values; take() masks them where they are consumed
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t ent = n0 + 2 * u + half;
      const int64_t er = ent < n ? ent : n - 1;
#pragma unroll
      for (int mt = 0; mt < NT; mt++) dst[u][mt] = X[er * k + min(mt * 32 + col, k - 1)];
    }
  };
  auto take = [&](const int64_t n0, const float (&src)[U][NT], float (&dst)[U][NT]) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const bool in = n0 + 2 * u + half < n;
#pragma unroll
      for (int mt = 0; mt < NT; mt++) dst[u][mt] = src[u][mt] * ((in && mt * 32 + col < k) ? 1.f : 0.f);
    }
  };
  float av[U][NT], nx[U][NT];
  const int64_t stride = nw * 2 * U;
  int64_t n0 = gw * 2 * U;
  if (n0 < n) {
    fetch(n0, nx);
    take(n0, nx, av);
  }
  for (; n0 < n; n0 += stride) {
    const bool more = n0 + stride < n;
    if (more) fetch(n0 + stride, nx);
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int mt = 0; mt < NT; mt++) amax = fmaxf(amax, fabsf(av[u][mt]));
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int mt = 0; mt < NT; mt++)
#pragma unroll
        for (int nt = 0; nt <= mt; nt++)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][mt], av[u][nt], acc[mt][nt], 0, 0, 0);
    if (more) take(n0 + stride, nx, av);
  }
  if (absmax_bits) {   // non-negative floats order like their bit patterns
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
    if (lane == 0) atomicMax(absmax_bits, __float_as_uint(amax < 3.0e38f ? amax : 3.0e38f));
  }
  float* out = partials + (size_t)gw * KP * KP;
#pragma unroll
  for (int mt = 0; mt < NT; mt++)
#pragma unroll
    for (int nt = 0; nt <= mt; nt++)
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int rr = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        out[(size_t)rr * KP + nt * 32 + col] = acc[mt][nt][e];
      }
}

__global__ void gramian_reduce_kernel(const float* __restrict__ partials, int nparts, int KP, int k, float ridge,
                                      float* __restrict__ G, double* __restrict__ diag) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= k * k) return;
  const int aa = e / k, bb = e % k;  // writes G[bb*k + aa]; bb fastest -> coalesced partial reads
  const int ta = aa / 32, tb = bb / 32;
  const int rr = ta >= tb ? aa : bb, cc = ta >= tb ? bb : aa;  // only tiles with row-tile >= col-tile exist
  double s = 0.0;
  const float* src = partials + (size_t)rr * KP + cc;
  for (int w = 0; w < nparts; w++) s += (double)src[(size_t)w * KP * KP];
  G[(size_t)bb * k + aa] = (float)s + (aa == bb ? ridge : 0.f);
  if (aa == bb) diag[aa] = s;
}

__global__ void trace_kernel(const double* __restrict__ diag, int k, double* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < k; i++) s += diag[i];
    *out = s;
  }
}

// sum_j w_j |X[:,j]|^2, deterministic: per-block partials then one block.
__global__ __launch_bounds__(256) void weighted_sumsq_kernel(const float* __restrict__ X, int k, int64_t n,
                                                             const float* __restrict__ w, double* __restrict__ partials) {
  __shared__ double sh[4];
  const int64_t total = n * k;
  double s = 0.0;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const float v = X[e];
    const float wj = w ? w[e / k] : 1.f;
    s += (double)(v * v * wj);
  }
  // wave reduce in double via shuffles
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ partials, size_t n,
                                                           double* __restrict__ out) {
  __shared__ double sh[4];
  double s = 0.0;
  for (size_t e = threadIdx.x; e < n; e += 256) s += partials[e];
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// first stage for long arrays: block b sums the b-th of gridDim.x equal contiguous chunks (fixed partition and order:
// deterministic) into tail[b]
__global__ __launch_bounds__(256) void sum_partials_stage_kernel(const double* __restrict__ partials, size_t n,
                                                                 double* __restrict__ tail) {
  __shared__ double sh[4];
  const size_t per = (n + gridDim.x - 1) / gridDim.x, lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  double s = 0.0;
  for (size_t e = lo + threadIdx.x; e < hi; e += 256) s += partials[e];
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) tail[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ void f64_to_f32_kernel(const double* __restrict__ in, float* __restrict__ out, size_t n) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x)
    out[e] = (float)in[e];
}
__global__ void f32_to_f64_kernel(const float* __restrict__ in, double* __restrict__ out, size_t n) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x)
    out[e] = (double)in[e];
}

template <typename K>
hipError_t set_lds(K kernel, size_t bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)bytes);
}

template <int KP, bool IMPLICIT, bool VEC>
hipError_t launch_cg_t(const AlsArgs& a, hipStream_t s, hipEvent_t* ev) {
  constexpr int T = kTileNnz, W = kWavesPerWG;
  using SM = CgSmem<KP, T, W, IMPLICIT>;
  hipError_t err;
  const int grid_s = (a.n_cols + kRowsPerWGShort - 1) / kRowsPerWGShort;
  if (ev && (err = hipEventRecord(ev[0], s)) != hipSuccess) return err;
  if (grid_s > 0) {
    auto ks = als_cg_short_kernel<KP, T, W, IMPLICIT, VEC>;
    if ((err = set_lds(ks, SM::short_bytes)) != hipSuccess) return err;
    prof_note(ev, reinterpret_cast<const void*>(ks));
    hipLaunchKernelGGL(ks, dim3(grid_s), dim3(W * 64), SM::short_bytes, s, a);
    if ((err = hipGetLastError()) != hipSuccess) return err;
  }
  if (ev && (err = hipEventRecord(ev[1], s)) != hipSuccess) return err;
  if (a.n_long > 0) {
    const int grid_l = (a.n_long + kRowsPerWGLong - 1) / kRowsPerWGLong;
    auto kl = als_cg_long_kernel<KP, T, W, IMPLICIT, VEC>;
    if ((err = set_lds(kl, SM::long_bytes)) != hipSuccess) return err;
    prof_note(ev ? ev + 1 : nullptr, reinterpret_cast<const void*>(kl));
    hipLaunchKernelGGL(kl, dim3(grid_l), dim3(W * 64), SM::long_bytes, s, a, (size_t)grid_s * W);
    if ((err = hipGetLastError()) != hipSuccess) return err;
  }
  if (ev && (err = hipEventRecord(ev[2], s)) != hipSuccess) return err;
  return hipSuccess;
}

}  // namespace

int padded_rank(int k) {
  if (k <= 0) return 0;
  if (k <= 32) return 32;
  if (k <= 64) return 64;
  if (k <= 128) return 128;
  return 0;
}

size_t cg_loss_slots(int n_cols, int n_long) {
  // one slot per short-kernel wave + one per long-kernel workgroup
  const size_t grid_s = ((size_t)n_cols + kRowsPerWGShort - 1) / kRowsPerWGShort;
  const size_t grid_l = ((size_t)n_long + kRowsPerWGLong - 1) / kRowsPerWGLong;
  return grid_s * kWavesPerWG + grid_l;
}

size_t chol_loss_slots(int n_cols) {  // = Cholesky grid: one slot per workgroup
  return (size_t)(n_cols < kCholMaxGrid ? (n_cols > 0 ? n_cols : 1) : kCholMaxGrid);
}

hipError_t launch_als_cg(const AlsArgs& a, bool implicit, hipStream_t s, hipEvent_t* ev) {
  const int KP = padded_rank(a.k);
  const bool vec = (a.k % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.X) & 15) == 0);
#define RSP_DISPATCH(KPV)                                                             \
  if (KP == KPV) {                                                                    \
    if (implicit) return vec ? launch_cg_t<KPV, true, true>(a, s, ev) : launch_cg_t<KPV, true, false>(a, s, ev); \
    return vec ? launch_cg_t<KPV, false, true>(a, s, ev) : launch_cg_t<KPV, false, false>(a, s, ev);             \
  }
  RSP_DISPATCH(32)
  RSP_DISPATCH(64)
  RSP_DISPATCH(128)
#undef RSP_DISPATCH
  return hipErrorInvalidValue;
}

// tail (nullable): kSumStageBlocks doubles of scratch; with it, long arrays are reduced in two stages
hipError_t launch_sum_partials(const double* partials, size_t n, double* out, hipStream_t s, double* tail) {
  if (tail && n > 16384) {
    hipLaunchKernelGGL(sum_partials_stage_kernel, dim3(kSumStageBlocks), dim3(256), 0, s, partials, n, tail);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, tail, (size_t)kSumStageBlocks, out);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, partials, n, out);
  return hipGetLastError();
}

static int gramian_waves(int64_t n) {
  int64_t w = (n + 63) / 64;  // >= 8 MFMA k-steps of work per wave
  if (w < 1) w = 1;
  if (w > 1024) w = 1024;
  return (int)((w + 3) / 4 * 4);
}

size_t gramian_scratch_floats(int k, int64_t n) {
  const int KP = padded_rank(k);
  return (size_t)gramian_waves(n) * KP * KP + 2 * 128 /* diag doubles */ + 16;
}

// absmax_bits (nullable): device word that receives max(its content, bits of max |X|) -- the caller zeroes it
hipError_t launch_gramian(const float* X, int k, int64_t n, float ridge, float* XtX, double* sumsq,
                          float* scratch, hipStream_t s, hipEvent_t* ev, unsigned* absmax_bits) {
  const int KP = padded_rank(k);
  if (!KP) return hipErrorInvalidValue;
  const int waves = gramian_waves(n);
  float* partials = scratch;
  double* diag = reinterpret_cast<double*>(scratch + (((size_t)waves * KP * KP + 1) & ~(size_t)1));
  const int grid = waves / 4;
  if (ev) (void)hipEventRecord(ev[0], s);
  prof_note(ev, KP == 32 ? reinterpret_cast<const void*>(gramian_partial_kernel<32>)
                         : (KP == 64 ? reinterpret_cast<const void*>(gramian_partial_kernel<64>)
                                     : reinterpret_cast<const void*>(gramian_partial_kernel<128>)));
  prof_note(ev ? ev + 1 : nullptr, reinterpret_cast<const void*>(gramian_reduce_kernel));
  if (KP == 32) hipLaunchKernelGGL(gramian_partial_kernel<32>, dim3(grid), dim3(256), 0, s, X, k, n, partials, absmax_bits);
  else if (KP == 64) hipLaunchKernelGGL(gramian_partial_kernel<64>, dim3(grid), dim3(256), 0, s, X, k, n, partials, absmax_bits);
  else hipLaunchKernelGGL(gramian_partial_kernel<128>, dim3(grid), dim3(256), 0, s, X, k, n, partials, absmax_bits);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return err;
  if (ev) (void)hipEventRecord(ev[1], s);
  hipLaunchKernelGGL(gramian_reduce_kernel, dim3((k * k + 255) / 256), dim3(256), 0, s, partials, waves, KP, k,
                     ridge, XtX, diag);
  if ((err = hipGetLastError()) != hipSuccess) return err;
  if (sumsq) {
    hipLaunchKernelGGL(trace_kernel, dim3(1), dim3(64), 0, s, diag, k, sumsq);
    err = hipGetLastError();
  }
  if (ev) (void)hipEventRecord(ev[2], s);
  return err;
}

hipError_t launch_weighted_sumsq(const float* X, int k, int64_t n, const float* w, double* out, double* scratch,
                                 hipStream_t s) {
  int64_t blocks = (n * k + 256 * 16 - 1) / (256 * 16);
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(weighted_sumsq_kernel, dim3((int)blocks), dim3(256), 0, s, X, k, n, w, scratch);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return err;
  return launch_sum_partials(scratch, (size_t)blocks, out, s);
}

hipError_t launch_f64_to_f32(const double* in, float* out, size_t n, hipStream_t s) {
  if (!n) return hipSuccess;
  size_t blocks = (n + 1023) / 1024;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(f64_to_f32_kernel, dim3((int)blocks), dim3(256), 0, s, in, out, n);
  return hipGetLastError();
}
hipError_t launch_f32_to_f64(const float* in, double* out, size_t n, hipStream_t s) {
  if (!n) return hipSuccess;
  size_t blocks = (n + 1023) / 1024;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(f32_to_f64_kernel, dim3((int)blocks), dim3(256), 0, s, in, out, n);
  return hipGetLastError();
}

}  // namespace rsparse_hip
