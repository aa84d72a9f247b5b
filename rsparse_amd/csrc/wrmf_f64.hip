// fp64 device path (gfx950): als_implicit<double> / als_explicit<double> -- what the reference's `*_double` entry points
// compute (src/wrmf_implicit.cpp:5-14, src/wrmf_explicit.cpp:5-14) and what `WRMF$new(precision = "double")`, the
// constructor's default (R/model_WRMF.R:82), runs.  One kernel family for every variant of the path: implicit / explicit
// feedback, Cholesky / conjugate gradient / NNLS, user/item biases, global bias.
//
// Per row (one workgroup, the row's k1 x k1 system in LDS):
//     lhs = XtX + X_nnz diag(c - 1) X_nnz^T            (wrmf_implicit.hpp:207-208)     | X_nnz X_nnz^T + lambda_use I
//     rhs = rhs_init + X_nnz (c - x_b % (c - 1))       (:226, :228-231)                | X_nnz (r - x_b)   (wrmf_explicit.hpp:89,103-106)
// assembled from the gathered factor vectors (chunks of the row staged in LDS) in 16 x 16 tiles of the lower triangle on
// v_mfma_f64_16x16x4_f64 (round 5; a wave a tile), then
//     Cholesky            right-looking LL^T in LDS in blocks of 8 columns: the diagonal tile factored in registers, a thread a
//                         row of the panel, the trailing update on the matrix cores (round 5); forward substitution riding
//                         along, backward substitution by one wave; a non-positive pivot sends the row -- assembled once more --
//                         to Gaussian elimination with partial pivoting in the same workgroup: what
//                         arma::solve(fast + likely_sympd) falls back to (:236, wrmf_explicit.hpp:108)
//     conjugate gradient  cg_solver_implicit / _global_bias / cg_solver_explicit (wrmf_implicit.hpp:8-57,
//                         wrmf_explicit.hpp:8-31) from the warm start, with A p evaluated from the assembled matrix: the same
//                         operator as XtX p + X_nnz((c-1) % X_nnz^T p), rounded differently at the 1e-16 level; the first
//                         residual of the global-bias variant is rhs - lhs x - g X_nnz (c - 1)
//     NNLS                c_nnls / scd_ls_update (nnls.hpp:10-48): XtX = lhs^T lhs + 1e-16 I, mu = XtX init - lhs^T rhs, the
//                         coordinate sweeps in order by one wave (lane = coordinate), stop at 1e-4 / 10000 sweeps
// and the loss term from a second pass over the row's vectors (L2-resident by then).
//
// This generic family is what every *_double entry point and WRMF(precision = "double") run for the exact solver, NNLS and the
// biased variants at every rank <= 128, and for the exact solve that ends every double fit (round 6: the class no longer
// falls back to fp32 above rank 63).  It moves k1^2 flops per non-zero (the fp32 conjugate-gradient kernels move
// 8 k (cg_steps + 1)) and one row occupies a workgroup; since round 5 its Cholesky is blocked (8-column panels, two barriers per
// block) with the trailing update and the assembly on v_mfma_f64_16x16x4_f64.  The bench line (fp32, BASELINE.json) does not
// run through it.
//
// The one configuration with kernels of its own is the constructor's default -- plain conjugate gradient, no biases
// (f64_cg_wave_kernel below, round 4; ranks 65..128 and the rows beyond 2048 non-zeros, f64_long_*_kernel, round 5): one wave
// per row (a wave per chunk of a long row), the operator applied from the gathered vectors as the reference does (no k1 x k1
// matrix).  Everything else goes through the workgroup-per-row family.
#include <algorithm>
#include <cstdlib>

#include <type_traits>
#include <utility>

#include "wrmf_f64.h"
#include "wrmf_internal.h"

namespace rsparse_hip {
namespace {

constexpr double kCgTolD = 1e-10;        // CG_TOL, inst/include/wrmf.hpp:22
template <class F, int... I>
__device__ __forceinline__ void static_for_f64_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for_f64(F&& f) {
  static_for_f64_impl(f, std::make_integer_sequence<int, N>{});
}

constexpr unsigned kScdMaxIter = 10000;  // SCD_MAX_ITER, wrmf.hpp:20
constexpr double kScdTol = 1e-4;         // SCD_TOL, wrmf.hpp:21
constexpr double kNnlsEps = 1e-16;       // EPS, nnls.hpp:8

__device__ __forceinline__ double wave_sum_d(double v) {   // butterfly: every lane ends with the same bits
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// v of lane `src` (wave-uniform) in every lane
__device__ __forceinline__ double f64_readlane_uniform(const double v, const int src) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}

// sum over the workgroup, every thread gets it; all threads must call it.  red: NT / 64 doubles
template <int NT>
__device__ __forceinline__ double block_sum(double v, double* red) {
  v = wave_sum_d(v);
  __syncthreads();   // red may still be read from the previous call
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < NT / 64; w++) s += red[w];
  return s;
}

// t-th 4 x 4 tile of the lower triangle, row-major over the tile rows: (ti, tj), tj <= ti
__device__ __forceinline__ void tile_of(int t, int& ti, int& tj) {
  ti = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
  while ((ti + 1) * (ti + 2) / 2 <= t) ti++;
  while (ti * (ti + 1) / 2 > t) ti--;
  tj = t - ti * (ti + 1) / 2;
}

// A[(4 ti + r) + (4 tj + c) lda] += sum_{j < cn} xs[j][4 ti + r] * w[j] * xs[j][4 tj + c] over the tiles of the lower triangle
template <int NT>
__device__ __forceinline__ void rank_update_tiles(double* A, int lda, const double* xs, int kp, const double* w, int cn,
                                                  int ntiles) {
  for (int t = threadIdx.x; t < ntiles; t += NT) {
    int ti, tj;
    tile_of(t, ti, tj);
    double acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int c = 0; c < 4; c++) acc[r][c] = 0.0;
    const double* xa = xs + 4 * ti;
    const double* xc = xs + 4 * tj;
    for (int j = 0; j < cn; j++) {
      const double wj = w ? w[j] : 1.0;
      double av[4], bv[4];
#pragma unroll
      for (int r = 0; r < 4; r++) av[r] = xa[j * kp + r];
#pragma unroll
      for (int c = 0; c < 4; c++) bv[c] = wj * xc[j * kp + c];
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) acc[r][c] = fma(av[r], bv[c], acc[r][c]);
    }
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int r = 0; r < 4; r++) A[(4 * ti + r) + (size_t)(4 * tj + c) * lda] += acc[r][c];
  }
}

using f64x4 = __attribute__((ext_vector_type(4))) double;

// The same update on the matrix cores: 16 x 16 tiles of the lower triangle (the whole diagonal tiles: their upper halves come out
// as the mirror image), one tile per wave and trip, four non-zeros per v_mfma_f64_16x16x4_f64 -- lane l hands over A[l & 15][l >> 4]
// and B[l >> 4][l & 15] and holds C/D[(l >> 4) + 4 r][l & 15], r < 4 (cdna_hip_programming.md: not the f32 map).  One instruction
// does the work of sixteen v_fma_f64; a workgroup of the generic kernel has its CU to itself (the matrix fills the LDS), so the
// instructions a row issues ARE its time (profiles/r05/r5_f64_generic_kernel_phases.txt).
// (The tile indices live in scalar registers -- the wave's number through readfirstlane --, and a tile that lies wholly inside the
//  matrix takes a path without masks: with one wave per SIMD the bookkeeping around the two to four matrix instructions of a tile,
//  not the instructions themselves, was its time -- 1800 cycles a tile, r5_f64_generic_kernel_phases.txt.)
template <int NT>
__device__ __forceinline__ void rank_update_mfma(double* A, int lda, int kp, const double* xs, const double* w, int cn) {
  constexpr int NW = NT / 64;
  const int lane = threadIdx.x & 63, r16 = lane & 15, q4 = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int t16 = (kp + 15) >> 4, cn4 = cn & ~3;
  int ti = 0, tj = wv;   // tiles (ti, tj), tj <= ti, row by row: this wave's first, then every NW-th
  while (tj > ti) {
    tj -= ti + 1;
    ti++;
  }
  while (ti < t16) {
    const int ra = 16 * ti + r16, rb = 16 * tj + r16;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    double* pc = A + (16 * ti + q4) + (size_t)rb * lda;
    if (16 * (ti + 1) <= kp) {   // (uniform) the whole tile exists
      const double* xa = xs + ra + q4 * kp;
      const double* xb = xs + rb + q4 * kp;
      const double* wq = w + q4;
      for (int kk = 0; kk < cn4; kk += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[kk * kp], wq[kk] * xb[kk * kp], acc, 0, 0, 0);
      if (cn4 < cn) {
        const bool vj = cn4 + q4 < cn;
        const int jo = vj ? cn4 : cn4 - 4 >= 0 ? cn4 - 4 : 0;   // (a lane beyond the chunk reads a slot that exists; its products are zeroed)
        const double av = xa[jo * kp], bv = wq[jo] * xb[jo * kp];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(vj ? av : 0.0, vj ? bv : 0.0, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; r++) pc[4 * r] += acc[r];
    } else {
      const bool va = ra < kp, vb = rb < kp;
      const double* xa = xs + min(ra, kp - 1);
      const double* xb = xs + min(rb, kp - 1);
      for (int kk = 0; kk < cn; kk += 4) {
        const int j = min(kk + q4, cn - 1);
        const bool vj = kk + q4 < cn;
        const double av = xa[j * kp], bv = w[j] * xb[j * kp];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64((va && vj) ? av : 0.0, (vb && vj) ? bv : 0.0, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; r++)
        if (16 * ti + q4 + 4 * r < kp && vb) pc[4 * r] += acc[r];
    }
    tj += NW;
    while (tj > ti) {
      tj -= ti + 1;
      ti++;
    }
  }
}

// (dev build -DRSP_F64_PROF, tools/gpu_f64_phases.sh: s_memtime ticks of workgroup 0 per phase of a row, printed at the end)
#ifdef RSP_F64_PROF
#define F64_T(j) { const unsigned long long _t1 = __builtin_amdgcn_s_memtime(); prof_t[j] += _t1 - _tl; _tl = _t1; }
#else
#define F64_T(j)
#endif

#ifndef RSP_F64_W64   // waves per SIMD asked of the one-wave instantiation (ranks <= 32): see the measurement at the launch
#define RSP_F64_W64 2
#endif
template <int NT>
__global__ __launch_bounds__(NT, NT == 64 ? RSP_F64_W64 : 2) void f64_als_kernel(F64Args a, int KP, int CH, int m2_in_lds) {   // (two waves per SIMD: two workgroups of 256 at rank 33..64, one of 512 beyond)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* sm = reinterpret_cast<double*>(smem_raw);
  constexpr int NW = NT / 64;
  const int LDA = KP + 1;   // odd: rows and columns of the matrix are both conflict-free walks
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int k = a.k, k1 = a.k1;
  const bool nnls = a.solver == 2, cg = a.solver == 1;
  double* A = sm;
  double* M2 = A + (size_t)KP * LDA;
  double* rhs = M2 + (m2_in_lds ? (size_t)KP * LDA : 0);
  if (nnls && !m2_in_lds) M2 = a.m2_scratch + (size_t)blockIdx.x * KP * LDA;
  double* x = rhs + KP;
  double* r = x + KP;
  double* p = r + KP;
  double* ap = p + KP;
  double* sv = ap + KP;     // X_nnz (c - 1): the global-bias term of the first CG residual
  double* invd = sv + KP;   // Cholesky: 1 / L_jj
  double* red = invd + 3 * (size_t)KP;   // 8 (two vectors' room unused: the general solver re-assembles the system it falls back on)
  double* xs = red + 8;     // [CH][KP] staged factor vectors of the current chunk
  double* cw = xs + (size_t)CH * KP;   // per staged non-zero: weight of x x^T
  double* rw = cw + CH;                // ... of x in the right-hand side
  double* lw = rw + CH;                // ... of the loss term
  double* lt = lw + CH;                // target of the loss term
  int* sidx = reinterpret_cast<int*>(lt + CH);
  int* spiv = sidx + CH;               // general solver: pivot row / singular flag

  const int T4 = KP / 4, ntiles = T4 * (T4 + 1) / 2;
  // trailing update of the factorisation: RT rows per pass (a power of two >= min(NT, k1)), NT / RT column groups
  int RT = 32;
  while (RT < k1 && RT < NT) RT <<= 1;
  const int CG = NT / RT, ri = tid & (RT - 1), cgi = tid / RT;

  double wloss = 0.0;   // thread 0: loss terms of this workgroup's rows, in row order
#ifdef RSP_F64_PROF
  unsigned long long prof_t[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long _tl = __builtin_amdgcn_s_memtime();
  int prof_rows = 0;
#endif

  auto stage = [&](const int p1, const int c0, const int cn) {
    if (tid < cn) {
      const int id = a.row_idx[p1 + c0 + tid];
      const double c = a.vals[p1 + c0 + tid];
      const double xbj = a.xb >= 0 ? a.X[(size_t)id * k + a.xb] : 0.0;
      sidx[tid] = id;
      if (a.implicit) {
        cw[tid] = c - 1.0;
        rw[tid] = c - xbj * (c - 1.0);                // wrmf_implicit.hpp:226 (x_b = 0: X_nnz c, :228-231)
        lw[tid] = c;
        lt[tid] = (1.0 - a.gbias) - xbj;              // :259-270
      } else {
        cw[tid] = 1.0;
        rw[tid] = c - xbj;                            // wrmf_explicit.hpp:89
        lw[tid] = 1.0;
        lt[tid] = c - xbj;                            // :131
      }
    }
    __syncthreads();
    // four vectors per wave and trip, all eight loads requested before the first is stored (the rank is at most 128: lane l
    // holds the coordinates l and 64 + l) -- a workgroup owns the CU here, nothing else hides a round trip
    for (int j0 = wv * 4; j0 < cn; j0 += NW * 4) {
      double v[4][2];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const double* base = a.X + (size_t)sidx[min(j0 + u, cn - 1)] * k + a.xoff;
        v[u][0] = base[min(lane, k1 - 1)];
        v[u][1] = base[min(lane + 64, k1 - 1)];
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (j0 + u < cn) {
          if (lane < KP) xs[(j0 + u) * KP + lane] = lane < k1 ? v[u][0] : 0.0;
          if (lane + 64 < KP) xs[(j0 + u) * KP + lane + 64] = lane + 64 < k1 ? v[u][1] : 0.0;
        }
    }
    __syncthreads();
  };
  // s = (A v)[t], A the full symmetric matrix
  auto matrow = [&](const double* M, const int t, const double* v) {
    double s0 = 0.0, s1 = 0.0;
    int c = 0;
    for (; c + 1 < k1; c += 2) {
      s0 = fma(M[t + (size_t)c * LDA], v[c], s0);
      s1 = fma(M[t + (size_t)(c + 1) * LDA], v[c + 1], s1);
    }
    if (c < k1) s0 = fma(M[t + (size_t)c * LDA], v[c], s0);
    return s0 + s1;
  };

  for (int row = blockIdx.x; row < a.n_cols; row += gridDim.x) {
    const int p1 = a.col_ptrs[row], n = a.col_ptrs[row + 1] - p1;
    double* yrow = a.Y + (size_t)row * k;
    if (n <= 0 && !a.solve_empty) {   // empty column -> zeros (wrmf_implicit.hpp:272-283, wrmf_explicit.hpp:133-144)
      for (int t = tid; t < k1; t += NT) yrow[a.ooff + t] = 0.0;
      continue;
    }
    const double lam_use = a.implicit ? a.lambda : a.lambda * (a.dynamic_lambda ? (double)n : 1.0);   // wrmf_explicit.hpp:78
    // the row's system into the LDS: lower triangle of A (+ whole diagonal tiles), rhs, sv
    auto assemble = [&]() {
      __syncthreads();   // the previous readers are done with the LDS
      F64_T(7)
      for (int i0 = 0; i0 < LDA; i0 += 64) {   // XtX (or zeros): eight columns per wave and trip, their loads in flight together
        const int i = i0 + lane;
        for (int c0 = wv * 8; c0 < KP; c0 += NW * 8) {
          double v[8];
#pragma unroll
          for (int u = 0; u < 8; u++) v[u] = a.implicit ? a.XtX[min(i, k1 - 1) + (size_t)min(c0 + u, k1 - 1) * k1] : 0.0;
#pragma unroll
          for (int u = 0; u < 8; u++)
            if (c0 + u < KP && i < LDA) A[i + (size_t)(c0 + u) * LDA] = (i < k1 && c0 + u < k1) ? v[u] : 0.0;
        }
      }
      for (int t = tid; t < KP; t += NT) {
        rhs[t] = (a.rhs_init && t < k1) ? a.rhs_init[t] : 0.0;
        sv[t] = 0.0;
        x[t] = t < k1 ? yrow[a.ioff + t] : 0.0;   // warm start (CG, NNLS): Y.col(i), drop_row(init, !is_x_bias_last_row)
      }
      __syncthreads();
      F64_T(0)
      for (int c0 = 0; c0 < n; c0 += CH) {
        const int cn = min(CH, n - c0);
        stage(p1, c0, cn);
        F64_T(12)
        rank_update_mfma<NT>(A, LDA, KP, xs, cw, cn);
        F64_T(13)
        for (int t = tid; t < k1; t += NT) {
          double s1 = 0.0, s2 = 0.0;
          for (int j = 0; j < cn; j++) {
            const double xv = xs[j * KP + t];
            s1 = fma(rw[j], xv, s1);
            s2 = fma(cw[j], xv, s2);
          }
          rhs[t] += s1;
          sv[t] += s2;
        }
        __syncthreads();
        F64_T(14)
      }
      F64_T(1)
      if (!a.implicit)
        for (int t = tid; t < k1; t += NT) A[t + (size_t)t * LDA] += lam_use;   // lhs.diag() += lambda_use
      __syncthreads();
    };
    // the upper triangle (conjugate gradient and NNLS multiply by the whole matrix, the general solver eliminates on it; the
    // Cholesky factorisation never looks there)
    auto mirror = [&]() {
      for (int c = wv; c < KP; c += NW)
        for (int i = lane; i < c; i += 64) A[i + (size_t)c * LDA] = A[c + (size_t)i * LDA];
      __syncthreads();
    };
    assemble();
    if (cg || nnls) mirror();

    F64_T(2)
    if (cg) {
      // ---- conjugate gradient on the assembled system (wrmf_implicit.hpp:8-57, wrmf_explicit.hpp:8-31) ----
      double part = 0.0;
      for (int t = tid; t < k1; t += NT) {
        const double rr = (rhs[t] - matrow(A, t, x)) - a.gbias * sv[t];
        r[t] = rr;
        p[t] = rr;
        part += rr * rr;
      }
      double rsold = block_sum<NT>(part, red);
      for (int it = 0; it < a.cg_steps; it++) {
        __syncthreads();
        part = 0.0;
        for (int t = tid; t < k1; t += NT) {
          const double s = matrow(A, t, p);
          ap[t] = s;
          part += p[t] * s;
        }
        const double alpha = rsold / block_sum<NT>(part, red);
        part = 0.0;
        for (int t = tid; t < k1; t += NT) {
          x[t] += alpha * p[t];
          const double rr = r[t] - alpha * ap[t];
          r[t] = rr;
          part += rr * rr;
        }
        const double rsnew = block_sum<NT>(part, red);
        if (rsnew < kCgTolD) break;
        const double beta = rsnew / rsold;
        for (int t = tid; t < k1; t += NT) p[t] = r[t] + p[t] * beta;
        rsold = rsnew;
      }
      __syncthreads();
    } else if (nnls) {
      // ---- c_nnls (nnls.hpp:36-48): XtX = lhs^T lhs + EPS I, mu = XtX init - lhs^T rhs ----
      for (int t = tid; t < ntiles; t += NT) {
        int ti, tj;
        tile_of(t, ti, tj);
        double acc[4][4];
#pragma unroll
        for (int rr = 0; rr < 4; rr++)
#pragma unroll
          for (int c = 0; c < 4; c++) acc[rr][c] = 0.0;
        for (int m = 0; m < k1; m++) {
          double av[4], bv[4];
#pragma unroll
          for (int rr = 0; rr < 4; rr++) av[rr] = A[(4 * ti + rr) + (size_t)m * LDA];
#pragma unroll
          for (int c = 0; c < 4; c++) bv[c] = A[(4 * tj + c) + (size_t)m * LDA];
#pragma unroll
          for (int rr = 0; rr < 4; rr++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc[rr][c] = fma(av[rr], bv[c], acc[rr][c]);
        }
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            const int gi = 4 * ti + rr, gc = 4 * tj + c;
            const double v = acc[rr][c] + (gi == gc ? kNnlsEps : 0.0);
            M2[gi + (size_t)gc * LDA] = v;
            M2[gc + (size_t)gi * LDA] = v;
          }
      }
      __syncthreads();
      for (int t = tid; t < k1; t += NT) p[t] = matrow(M2, t, x) - matrow(A, t, rhs);   // mu
      __syncthreads();
      if (wv == 0) {
        // scd_ls_update (nnls.hpp:10-34): lane l owns coordinates l and l + 64
        const bool own0 = lane < k1, own1 = lane + 64 < k1;
        double h0 = own0 ? x[lane] : 0.0, h1 = own1 ? x[lane + 64] : 0.0;
        double mu0 = own0 ? p[lane] : 0.0, mu1 = own1 ? p[lane + 64] : 0.0;
        const double dg0 = own0 ? M2[lane + (size_t)lane * LDA] : 1.0;
        const double dg1 = own1 ? M2[(lane + 64) + (size_t)(lane + 64) * LDA] : 1.0;
        // (a coordinate at its bound with a non-negative gradient does not move -- new = max(0, 0 - mu / d) = 0 -- and the
        // reference's loop body does nothing for it, nnls.hpp:24: the sweep visits only the lanes whose coordinate can move,
        // re-evaluated after every move; the same sequence of updates without the idle visits, three quarters of all visits)
        for (unsigned it = 0; it < kScdMaxIter; it++) {
          double rel = 0.0;
          for (int half = 0; half < 2; half++) {
            const bool hi = half == 1;
            const int lim = min(64, k1 - 64 * half);
            if (lim <= 0) break;
            const unsigned long long in_range = lim >= 64 ? ~0ull : ((1ull << lim) - 1ull);
            unsigned long long act = __ballot(!((hi ? h1 : h0) == 0.0 && (hi ? mu1 : mu0) >= 0.0)) & in_range;
            while (act) {
              const int src = __builtin_ctzll(act);
              const int c = 64 * half + src;
              const double old = __shfl(hi ? h1 : h0, src);
              const double muc = __shfl(hi ? mu1 : mu0, src);
              const double dg = __shfl(hi ? dg1 : dg0, src);
              double nv = old - muc / dg;
              if (nv < 0.0) nv = 0.0;
              const double diff = nv - old;
              const unsigned long long above = src >= 63 ? 0ull : (~0ull << (src + 1));
              if (diff != 0.0) {
                if (lane == src) {
                  if (hi) h1 = nv; else h0 = nv;
                }
                const double* col = M2 + (size_t)c * LDA;
                if (own0) mu0 += diff * col[lane];
                if (own1) mu1 += diff * col[lane + 64];
                const double se = fabs(diff) / (fabs(old) + kNnlsEps);
                if (se > rel) rel = se;
                act = __ballot(!((hi ? h1 : h0) == 0.0 && (hi ? mu1 : mu0) >= 0.0)) & in_range & above;
              } else {
                act &= above;
              }
            }
          }
          if (rel <= kScdTol) break;
        }
        if (own0) x[lane] = h0;
        if (own1) x[lane + 64] = h1;
      }
      __syncthreads();
    } else {
      // ---- Cholesky, right-looking in blocks of 8 columns; z = L^-1 rhs rides along in x ----
      // (Column by column -- two barriers and a rank-1 sweep of the LDS per column -- the factorisation was 300 of the 360 us a
      // rank-128 row took, and WRMF's closing exact solve of 1M users 1.4 s: profiles/r05/r5q_*.  A block's 8 x 8 diagonal tile is
      // factored by EVERY thread for itself, in registers -- the same bits everywhere, no exchange --, then a thread solves its own row
      // of the panel against it and the trailing matrix takes the block's eight rank-1 terms in one sweep of 4 x 4 register tiles:
      // two barriers per block.)
      bool ok = true;
      constexpr int NB = 8;
      // one block of columns; FULL: all eight exist (every block but the last of a rank that is no multiple of 8)
      auto block = [&](auto full_t, const int jb) {
        constexpr bool FULL = decltype(full_t)::value;
        const int nb = FULL ? NB : k1 - jb, j2 = jb + nb;
        // the tile is factored by the waves that own rows of the panel (and wave 0, which writes the factor back): two waves of a
        // SIMD factoring the same tile side by side took twice as long for nothing (eight waves: ranks beyond 64)
        double L[NB][NB], dinv[NB], z[NB];
        if (wv == 0 || wv * 64 < k1 - j2) {
#pragma unroll
          for (int c = 0; c < NB; c++)
#pragma unroll
            for (int rr = c; rr < NB; rr++)   // (columns beyond the matrix: an identity tail keeps the loops whole)
              L[rr][c] = (FULL || (rr < nb && c < nb)) ? A[(jb + min(rr, nb - 1)) + (size_t)(jb + min(c, nb - 1)) * LDA] : (rr == c ? 1.0 : 0.0);
#pragma unroll
          for (int c = 0; c < NB; c++) z[c] = (FULL || c < nb) ? rhs[jb + min(c, nb - 1)] : 0.0;
          bool bad = false;
#pragma unroll
          for (int c = 0; c < NB; c++) {
            const double d = L[c][c];
            bad = bad || !(d > 0.0);
            const double di = rsqrt(d);   // (one reciprocal square root on the pivot chain instead of a square root and a division: 128 of them in a row)
            dinv[c] = di;
#pragma unroll
            for (int rr = c + 1; rr < NB; rr++) L[rr][c] *= di;
#pragma unroll
            for (int c2 = c + 1; c2 < NB; c2++)
#pragma unroll
              for (int rr = c2; rr < NB; rr++) L[rr][c2] = fma(-L[rr][c], L[c2][c], L[rr][c2]);
            double zc = z[c];
#pragma unroll
            for (int q = 0; q < c; q++) zc = fma(-L[c][q], z[q], zc);
            z[c] = zc * di;
          }
          if (tid == 0) spiv[1] = bad ? 1 : 0;   // (every wave that factors sees the same tile; the others read the verdict)
          F64_T(8)
          if (!bad)
            for (int i = j2 + tid; i < k1; i += NT) {   // the panel: row i of L against the tile, in registers
              double lr[NB], av[NB];
              double bi = rhs[i];
#pragma unroll
              for (int c = 0; c < NB; c++) av[c] = (FULL || c < nb) ? A[i + (size_t)(jb + min(c, nb - 1)) * LDA] : 0.0;
#pragma unroll
              for (int c = 0; c < NB; c++) {
                double v = av[c];
#pragma unroll
                for (int q = 0; q < c; q++) v = fma(-lr[q], L[c][q], v);
                lr[c] = v * dinv[c];
                if (FULL || c < nb) A[i + (size_t)(jb + c) * LDA] = lr[c];
                bi = fma(-lr[c], z[c], bi);
              }
              rhs[i] = bi;
            }
        }
        __syncthreads();   // the panel is in place (and nobody reads the diagonal tile any more)
        F64_T(9)
        if (spiv[1]) return false;
        if (tid == 0) {   // the tile's factor
#pragma unroll
          for (int c = 0; c < NB; c++)
            if (FULL || c < nb) {
              invd[jb + c] = dinv[c];
              x[jb + c] = z[c];
#pragma unroll
              for (int rr = c + 1; rr < NB; rr++)
                if (FULL || rr < nb) A[(jb + rr) + (size_t)(jb + c) * LDA] = L[rr][c];
            }
        }
        if (FULL && j2 < k1) {
          // trailing matrix -= panel panel^T on the matrix cores: 16 x 16 tiles from (j2, j2) on, two instructions each (K = 8);
          // the last tile row hangs over the matrix when KP - j2 is no multiple of 16: those tiles take the masked path
          const int r16 = lane & 15, q4 = lane >> 4;
          const int wvs = __builtin_amdgcn_readfirstlane(wv);
          const int t16 = (KP - j2 + 15) >> 4;
          const double* pan = A + (size_t)(jb + q4) * LDA;   // column jb + q4 of the panel (the second instruction: four columns on)
          int ti = 0, tj = wvs;
          while (tj > ti) {
            tj -= ti + 1;
            ti++;
          }
          while (ti < t16) {
            const int ra = j2 + 16 * ti + r16, rb = j2 + 16 * tj + r16;
            f64x4 cc;
            if (j2 + 16 * (ti + 1) <= KP) {   // (uniform)
              double* pc = A + (j2 + 16 * ti + q4) + (size_t)rb * LDA;
              const double a0 = pan[ra], a1 = pan[ra + 4 * (size_t)LDA], b0v = pan[rb], b1v = pan[rb + 4 * (size_t)LDA];
#pragma unroll
              for (int r = 0; r < 4; r++) cc[r] = pc[4 * r];
              cc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0, b0v, cc, 0, 0, 0);
              cc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1, b1v, cc, 0, 0, 0);
#pragma unroll
              for (int r = 0; r < 4; r++) pc[4 * r] = cc[r];
            } else {
              const int rac = min(ra, KP - 1), rbc = min(rb, KP - 1);
              double* pc = A + (j2 + 16 * ti + q4) + (size_t)rbc * LDA;
              const double a0 = pan[rac], a1 = pan[rac + 4 * (size_t)LDA], b0v = pan[rbc], b1v = pan[rbc + 4 * (size_t)LDA];
              bool vc[4];
#pragma unroll
              for (int r = 0; r < 4; r++) {
                vc[r] = j2 + 16 * ti + q4 + 4 * r < KP && rb < KP;
                cc[r] = vc[r] ? pc[4 * r] : 0.0;
              }
              cc = __builtin_amdgcn_mfma_f64_16x16x4f64(ra < KP ? -a0 : 0.0, rb < KP ? b0v : 0.0, cc, 0, 0, 0);
              cc = __builtin_amdgcn_mfma_f64_16x16x4f64(ra < KP ? -a1 : 0.0, rb < KP ? b1v : 0.0, cc, 0, 0, 0);
#pragma unroll
              for (int r = 0; r < 4; r++)
                if (vc[r]) pc[4 * r] = cc[r];
            }
            tj += NW;
            while (tj > ti) {
              tj -= ti + 1;
              ti++;
            }
          }
        }
        F64_T(10)
        return true;
      };
      for (int jb = 0; jb < k1 && ok; jb += NB) {
        __syncthreads();   // the previous block's trailing update has landed
        F64_T(11)
        ok = k1 - jb >= NB ? block(std::true_type{}, jb) : block(std::false_type{}, jb);
      }
      __syncthreads();
      F64_T(3)
      if (ok) {
        if (wv == 0) {   // L^T y = z: lane l holds entries l and l + 64
          double z0 = lane < k1 ? x[lane] : 0.0, z1 = lane + 64 < k1 ? x[lane + 64] : 0.0;
          // 1 / L_mm and row m of L wait in registers (the row a step ahead, zeroed beyond the diagonal), and a lane's entry is final
          // once its own step has passed (y_m = z_m / L_mm, formed at the end as in the step): the chain of a step is a product, a
          // lane read and an fma -- it was two LDS round trips, two selects and a compare as well
          const double i0 = lane < k1 ? invd[lane] : 0.0, i1 = lane + 64 < k1 ? invd[lane + 64] : 0.0;
          const int c0l = min(lane, KP - 1), c1l = min(lane + 64, KP - 1);
          // (Rows four steps ahead, in rings of registers, were slower: 1.43 -> 1.82 M ticks per 82 rows at rank 128.)
          int m = k1 - 1;
          double a0 = A[m + (size_t)c0l * LDA], a1 = A[m + (size_t)c1l * LDA];
          a1 = lane + 64 < m ? a1 : 0.0;
          for (; m >= 64; m--) {   // (lane < m throughout)
            const int mn = m - 1;
            const double n0 = A[mn + (size_t)c0l * LDA], n1r = A[mn + (size_t)c1l * LDA];
            const double n1 = lane + 64 < mn ? n1r : 0.0;
            const double ym = f64_readlane_uniform(z1 * i1, m - 64);
            z0 = fma(-a0, ym, z0);
            z1 = fma(-a1, ym, z1);
            a0 = n0;
            a1 = n1;
          }
          a0 = lane < m ? a0 : 0.0;
          for (; m >= 0; m--) {
            const int mn = max(m - 1, 0);
            const double n0r = A[mn + (size_t)c0l * LDA];
            const double n0 = lane < mn ? n0r : 0.0;
            const double ym = f64_readlane_uniform(z0 * i0, m);
            z0 = fma(-a0, ym, z0);
            a0 = n0;
          }
          if (lane < k1) x[lane] = z0 * i0;
          if (lane + 64 < k1) x[lane + 64] = z1 * i1;
        }
        __syncthreads();
      } else {
        // ---- the general solver (gesv's order: partial pivoting, first largest entry): the system once more, whole ----
        assemble();
        mirror();
        bool singular = false;
        for (int c = 0; c < k1; c++) {
          __syncthreads();
          if (wv == 0) {
            double best = -1.0;
            int bi = c;
            for (int i = c + lane; i < k1; i += 64) {
              const double v = fabs(A[i + (size_t)c * LDA]);
              if (v > best) { best = v; bi = i; }
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
              const double ov = __shfl_xor(best, m);
              const int oi = __shfl_xor(bi, m);
              if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            if (lane == 0) {
              spiv[0] = bi;
              spiv[1] = best > 0.0 ? 0 : 1;
            }
          }
          __syncthreads();
          if (spiv[1]) {
            singular = true;
            break;
          }
          const int piv = spiv[0];
          if (piv != c) {
            for (int m = tid; m < k1; m += NT) {
              const double t0 = A[c + (size_t)m * LDA];
              A[c + (size_t)m * LDA] = A[piv + (size_t)m * LDA];
              A[piv + (size_t)m * LDA] = t0;
            }
            if (tid == 0) {
              const double t0 = rhs[c];
              rhs[c] = rhs[piv];
              rhs[piv] = t0;
            }
          }
          __syncthreads();
          const double pinv = 1.0 / A[c + (size_t)c * LDA];
          const double bc = rhs[c];
          for (int i = c + 1 + ri; i < k1; i += RT) {
            const double f = A[i + (size_t)c * LDA] * pinv;
            if (f != 0.0) {
              for (int m = c + 1 + cgi; m < k1; m += CG) A[i + (size_t)m * LDA] = fma(-f, A[c + (size_t)m * LDA], A[i + (size_t)m * LDA]);
              if (cgi == 0) rhs[i] = fma(-f, bc, rhs[i]);
            }
          }
        }
        __syncthreads();
        if (!singular) {
          if (wv == 0) {   // U y = b: lane l holds entries l and l + 64
            double z0 = lane < k1 ? rhs[lane] : 0.0, z1 = lane + 64 < k1 ? rhs[lane + 64] : 0.0;
            for (int m = k1 - 1; m >= 0; m--) {
              const int src = m & 63;
              const bool hi = m >= 64;
              const double ym = f64_readlane_uniform(hi ? z1 : z0, src) / A[m + (size_t)m * LDA];
              if (lane == src) {
                if (hi) z1 = ym; else z0 = ym;
              }
              if (lane < m) z0 = fma(-A[lane + (size_t)m * LDA], ym, z0);
              if (lane + 64 < m) z1 = fma(-A[(lane + 64) + (size_t)m * LDA], ym, z1);
            }
            if (lane < k1) x[lane] = z0;
            if (lane + 64 < k1) x[lane + 64] = z1;
          }
        } else {
          for (int t = tid; t < k1; t += NT) x[t] = 0.0;
        }
        if (tid == 0 && a.fail_counter) {
          atomicAdd(a.fail_counter + 2, 1);
          if (singular) atomicAdd(a.fail_counter + 3, 1);
        }
        __syncthreads();
      }
    }

    F64_T(4)
    // ---- write back, loss term (wrmf_implicit.hpp:254-270, wrmf_explicit.hpp:113-132) ----
    for (int t = tid; t < k1; t += NT) yrow[a.ooff + t] = x[t];
    // (the row's vectors once more, straight from memory: a wave takes four non-zeros per trip, no staging, no barrier)
    double lpart = 0.0;
    {
      const double x0 = lane < k1 ? x[lane] : 0.0, x1 = lane + 64 < k1 ? x[lane + 64] : 0.0;
      for (int j0 = wv * 4; j0 < n; j0 += NW * 4) {
        double v[4][2], cj[4], xbj[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int jj = p1 + min(j0 + u, n - 1);
          const double* col = a.X + (size_t)a.row_idx[jj] * k;
          cj[u] = a.vals[jj];
          v[u][0] = col[a.xoff + min(lane, k1 - 1)];
          v[u][1] = col[a.xoff + min(lane + 64, k1 - 1)];
          xbj[u] = a.xb >= 0 ? col[a.xb] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          double sd = lane < k1 ? v[u][0] * x0 : 0.0;
          sd = lane + 64 < k1 ? fma(v[u][1], x1, sd) : sd;
          sd = wave_sum_d(sd);
          const double lwj = a.implicit ? cj[u] : 1.0;                                    // (as `stage` forms them)
          const double ltj = a.implicit ? (1.0 - a.gbias) - xbj[u] : cj[u] - xbj[u];   // wrmf_implicit.hpp:259-270, wrmf_explicit.hpp:131
          const double dlt = ltj - sd;
          lpart += j0 + u < n ? lwj * dlt * dlt : 0.0;
        }
      }
    }
    double yy = 0.0;
    for (int t = tid; t < k1; t += NT) yy += x[t] * x[t];
    const double tot = block_sum<NT>((lane == 0 ? lpart : 0.0) + lam_use * yy, red);
    if (tid == 0) wloss += tot;
    F64_T(5)
#ifdef RSP_F64_PROF
    prof_rows++;
#endif
  }
#ifdef RSP_F64_PROF
  if (blockIdx.x == 0 && tid == 0)
    printf("f64_als_kernel phases (ticks of s_memtime, workgroup 0, %d rows): copy %llu assembly %llu [stage %llu update %llu rhs %llu] mirror %llu solve %llu [tile %llu panel %llu trailing %llu barrier %llu] backsub %llu loss %llu between %llu\n",
           prof_rows, prof_t[0], prof_t[1] + prof_t[12] + prof_t[13] + prof_t[14], prof_t[12], prof_t[13], prof_t[14], prof_t[2],
           prof_t[3] + prof_t[8] + prof_t[9] + prof_t[10] + prof_t[11], prof_t[8], prof_t[9], prof_t[10], prof_t[11], prof_t[4], prof_t[5], prof_t[7]);
#endif
  if (tid == 0) a.loss_partials[blockIdx.x] = wloss;
}

// ---- the plain conjugate-gradient half-iteration, one WAVE per row (end of round 4) -------------------------------------------
// cg_solver_implicit / cg_solver_explicit (wrmf_implicit.hpp:8-32, wrmf_explicit.hpp:8-31) with the operator evaluated as the
// reference evaluates it -- A v = XtX v + X_nnz ((c - 1) o (X_nnz^T v)) (explicit: X_nnz X_nnz^T v + lambda_use v) -- instead of on
// an assembled k x k system: the reference's DEFAULT configuration (rank 10, implicit, conjugate gradient, precision "double") ran
// at 6 of 64 lanes in the generic kernel above (its 4 x 4 assembly tiles: six of them at rank 10..12) -- 53 ms per half-iteration
// at 1M x 100k where the fp32 path takes 4.  No bias operands, no global bias; everything else stays with the generic kernel.
// Layout: a group of W lanes (16 / 32 / 64: the rank rounded up) holds one gathered vector, lane l of the group = coordinate l, so a
// step takes 64 / W non-zeros; x, r, p, A p live one coordinate per lane, replicated in every group.
// lane-crossing helpers on doubles (two 32-bit halves through the DPP / lane-swap paths: no LDS round trips)
template <int CTRL>
__device__ __forceinline__ double f64_dpp(const double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int W>
__device__ __forceinline__ double f64_group_sum(double v) {   // every lane of a group of W (16 / 32 / 64) ends with the group's sum
  v += f64_dpp<0xB1>(v);    // quad_perm:[1,0,3,2]
  v += f64_dpp<0x4E>(v);    // quad_perm:[2,3,0,1]
  v += f64_dpp<0x141>(v);   // row_half_mirror
  v += f64_dpp<0x140>(v);   // row_mirror: every lane of a row of 16 holds the row's sum
  if constexpr (W >= 32) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto sl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);   // rows (0, 0, 2, 2) and (1, 1, 3, 3)
    const auto sh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __hiloint2double((int)sh[0], (int)sl[0]) + __hiloint2double((int)sh[1], (int)sl[1]);
  }
  if constexpr (W == 64) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto sl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);   // halves (0, 0) and (1, 1)
    const auto sh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    v = __hiloint2double((int)sh[0], (int)sl[0]) + __hiloint2double((int)sh[1], (int)sl[1]);
  }
  return v;
}
// the sum over the groups of a value held per lane (same coordinate in every group)
template <int W>
__device__ __forceinline__ double f64_across_groups(double v) {
  if constexpr (W == 16) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto sl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto sh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __hiloint2double((int)sh[0], (int)sl[0]) + __hiloint2double((int)sh[1], (int)sl[1]);
  }
  if constexpr (W <= 32) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto sl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto sh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    v = __hiloint2double((int)sh[0], (int)sl[0]) + __hiloint2double((int)sh[1], (int)sl[1]);
  }
  return v;
}
// the value lane (group g, position ST) of the wave holds, in every lane of group g
template <int W, int ST>
__device__ __forceinline__ int f64_from_step_lane(const int v, const int g) {
  if constexpr (W == 16) {
    return __builtin_amdgcn_update_dpp(0, v, 0x150 + ST, 0xf, 0xf, true);   // row_newbcast:ST
  } else if constexpr (W == 32) {
    const int a0 = __builtin_amdgcn_readlane(v, ST), a1 = __builtin_amdgcn_readlane(v, 32 + ST);
    return g ? a1 : a0;
  } else {
    return __builtin_amdgcn_readlane(v, ST);
  }
}

// Sixteen per-lane doubles p[0..15] summed over the 64 lanes, TRANSPOSED: lane L ends with the wave's sum of p[L >> 2] (the four
// lanes of a quad hold copies).  Every stage halves the number of values a lane carries: a lane swap fed two DIFFERENT registers
// is the exchange step of both (lanes of the low half keep the first one's sum, the high half the second's), then the same inside
// the rows of 16 lanes with a select and one DPP move.  15 additions and 15 exchanges per lane for 16 sums, where sixteen
// butterflies (f64_group_sum) take 96 of each -- the lane-crossing sum of a non-zero's dot product was most of the fp64
// conjugate-gradient kernel's instructions at rank 65..128 (round 5).
__device__ __forceinline__ double f64_transposed_sum16(double (&p)[16], const int lane) {
  auto swap32 = [](const double a, const double b) {
    const auto sl = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto sh = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)sh[0], (int)sl[0]) + __hiloint2double((int)sh[1], (int)sl[1]);
  };
  auto swap16 = [](const double a, const double b) {
    const auto sl = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto sh = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)sh[0], (int)sl[0]) + __hiloint2double((int)sh[1], (int)sl[1]);
  };
  double s8[8], s4[4], s2[2];
#pragma unroll
  for (int u = 0; u < 8; u++) s8[u] = swap32(p[u], p[u + 8]);       // half h: non-zero u + 8 h
#pragma unroll
  for (int u = 0; u < 4; u++) s4[u] = swap16(s8[u], s8[u + 4]);     // row r: non-zero u + 4 (r & 1) + 8 (r >> 1)
  const bool b8 = (lane & 8) != 0, b4 = (lane & 4) != 0;
#pragma unroll
  for (int u = 0; u < 2; u++) {                                     // lanes with bit 8: u + 2
    const double keep = b8 ? s4[u + 2] : s4[u], send = b8 ? s4[u] : s4[u + 2];
    s2[u] = keep + f64_dpp<0x128>(send);                            // row_ror:8 = the lane 8 away inside the row
  }
  const double keep = b4 ? s2[1] : s2[0], send = b4 ? s2[0] : s2[1];
  double t = keep + f64_dpp<0x141>(send);                           // row_half_mirror: lane i <-> 7 - i of its half row
  t += f64_dpp<0xB1>(t);                                            // quad_perm:[1,0,3,2]
  t += f64_dpp<0x4E>(t);                                            // quad_perm:[2,3,0,1]
  return t;
}

template <int EPL> struct F64Vec { double c[EPL]; };

// One pass over n non-zeros at p1 (a row, or a chunk of a long row): out_l = sum_j coef(c_j, x_j . v) x_j[l], or (MODE 2) sum_j lw_j (lt_j - x_j . v)^2 in
// every lane.  MODE 0: first residual, 1: operator, 2: loss.  A chunk is 64 non-zeros: lane (g, st) holds the index and the value of the
// non-zero st * NPS + g, the one group g gathers in step st.  lk / lc: the lane's coordinates (valid, clamped), l / g: lane within its group, group.
template <int W, bool IMPLICIT, int EPL, int MODE>
__device__ __forceinline__ F64Vec<EPL> f64_row_pass(const F64Args& a, const int k, const int p1, const int n, const F64Vec<EPL> v,
                                                    const bool (&lk)[EPL], const int (&lc)[EPL], const int l, const int g, const int lane) {
  using Vec = F64Vec<EPL>;
  constexpr int NPS = 64 / W;   // non-zeros per step
  constexpr int BS = 16;        // steps per batch: their vectors are requested together, the next batch's before this one is used
  constexpr int NBATCH = W / BS;
  Vec acc;
#pragma unroll
  for (int e = 0; e < EPL; e++) acc.c[e] = 0.0;
  // (index, value) of the lane's non-zero of the chunk at c0, requested two chunks ahead of its use; the vectors of a batch of
  // 16 steps one batch ahead, across chunk boundaries: a row of 500 non-zeros is 8 chunks x 5 passes, and every chunk used to
  // cost two exposed round trips (its indices, then its vectors)
  auto meta = [&](const int c0, int& idj, int& cjl, int& cjh) {
    if (c0 < n) {   // wave-uniform
      const int mine = min(l * NPS + g, n - c0 - 1);
      idj = a.row_idx[p1 + c0 + mine];
      const double cj = a.vals[p1 + c0 + mine];
      cjl = __double2loint(cj);
      cjh = __double2hiint(cj);
    }
  };
  auto fetch = [&](auto bt, const int idj, const int nst, double (&dst)[BS][EPL]) {
    constexpr int B = decltype(bt)::value;
    if (B * BS < nst) {   // wave-uniform
      static_for_f64<BS>([&](auto ut) {
        constexpr int U = decltype(ut)::value, ST = B * BS + U;
        const int id = f64_from_step_lane<W, ST>(idj, g);   // (steps beyond the chunk repeat its last non-zero: weight 0)
#pragma unroll
        for (int e = 0; e < EPL; e++) dst[U][e] = a.X[(size_t)id * k + lc[e]];
      });
    }
  };
  int id0 = 0, c0l = 0, c0h = 0, id1 = 0, c1l = 0, c1h = 0, id2 = 0, c2l = 0, c2h = 0;
  meta(0, id0, c0l, c0h);
  meta(64, id1, c1l, c1h);
  double cur[BS][EPL], nxt[BS][EPL];
  fetch(std::integral_constant<int, 0>{}, id0, (min(64, n) + NPS - 1) / NPS, cur);
  for (int c0 = 0; c0 < n; c0 += 64) {
    const int cn = min(64, n - c0);
    const int nst = (cn + NPS - 1) / NPS;                                       // steps of this chunk (<= W)
    const int nst1 = c0 + 64 < n ? (min(64, n - c0 - 64) + NPS - 1) / NPS : 0;   // ... of the next one
    meta(c0 + 128, id2, c2l, c2h);
    static_for_f64<NBATCH>([&](auto bt) {
      constexpr int B = decltype(bt)::value;
      if (B * BS < nst) {   // wave-uniform
        // the batch after this one: of this chunk, or the first of the next chunk
        if ((B + 1) * BS < nst) {
          if constexpr (B + 1 < NBATCH) fetch(std::integral_constant<int, (B + 1 < NBATCH ? B + 1 : 0)>{}, id0, nst, nxt);
        } else if (nst1 > 0) {
          fetch(std::integral_constant<int, 0>{}, id1, nst1, nxt);
        }
        if constexpr (W == 64) {
          // ranks 33..128 (the whole wave holds one vector): the batch's sixteen dot products by ONE transposed reduction
          // (lane L: non-zero L >> 2 of the batch), the coefficients formed sixteen at a time, handed back lane by lane for
          // the sum over the vectors.  (Ranks 33..64 kept the per-non-zero butterfly -- six fp64 exchange stages each --
          // until the end of round 5, and a rank-64 fit took as long as a rank-128 one: profiles/r05/r5z_f64_per_iteration.txt.)
          double part[BS];
          static_for_f64<BS>([&](auto ut) {
            constexpr int U = decltype(ut)::value;
            part[U] = (lk[0] ? cur[U][0] : 0.0) * v.c[0];
#pragma unroll
            for (int e = 1; e < EPL; e++) part[U] = fma(lk[e] ? cur[U][e] : 0.0, v.c[e], part[U]);
          });
          const double t = f64_transposed_sum16(part, lane);
          const int st = B * BS + (lane >> 2);                        // the lane's non-zero of the chunk
          const double c = __hiloint2double(__shfl(c0h, st), __shfl(c0l, st));
          const bool in = st < cn;
          if constexpr (MODE == 2) {
            const double dlt = (IMPLICIT ? 1.0 : c) - t;
            acc.c[0] += (in && (lane & 3) == 0) ? (IMPLICIT ? c : 1.0) * dlt * dlt : 0.0;   // (one lane of the quad counts it)
          } else {
            double coef;
            if constexpr (MODE == 0) coef = IMPLICIT ? c - (c - 1.0) * t : c - t;
            else coef = IMPLICIT ? (c - 1.0) * t : t;
            coef = in ? coef : 0.0;
            const int ch = __double2hiint(coef), cl = __double2loint(coef);
            static_for_f64<BS>([&](auto ut) {
              constexpr int U = decltype(ut)::value, ST = B * BS + U;
              if (ST < nst) {   // wave-uniform
                const double cu = __hiloint2double(__builtin_amdgcn_readlane(ch, 4 * U), __builtin_amdgcn_readlane(cl, 4 * U));
#pragma unroll
                for (int e = 0; e < EPL; e++) acc.c[e] = fma(cu, lk[e] ? cur[U][e] : 0.0, acc.c[e]);
              }
            });
          }
        } else
        static_for_f64<BS>([&](auto ut) {
          constexpr int U = decltype(ut)::value, ST = B * BS + U;
          if (ST < nst) {   // wave-uniform
            double yv[EPL];
            double part = 0.0;
#pragma unroll
            for (int e = 0; e < EPL; e++) {
              yv[e] = lk[e] ? cur[U][e] : 0.0;
              part = fma(yv[e], v.c[e], part);
            }
            const bool in = ST * NPS + g < cn;
            const double c = __hiloint2double(f64_from_step_lane<W, ST>(c0h, g), f64_from_step_lane<W, ST>(c0l, g));
            const double t = f64_group_sum<W>(part);
            if constexpr (MODE == 2) {
              const double dlt = (IMPLICIT ? 1.0 : c) - t;
              acc.c[0] += in ? (IMPLICIT ? c : 1.0) * dlt * dlt : 0.0;
            } else {
              double coef;
              if constexpr (MODE == 0) coef = IMPLICIT ? c - (c - 1.0) * t : c - t;
              else coef = IMPLICIT ? (c - 1.0) * t : t;
              coef = in ? coef : 0.0;
#pragma unroll
              for (int e = 0; e < EPL; e++) acc.c[e] = fma(coef, yv[e], acc.c[e]);
            }
          }
        });
#pragma unroll
        for (int u = 0; u < BS; u++)
#pragma unroll
          for (int e = 0; e < EPL; e++) cur[u][e] = nxt[u][e];
      }
    });
    id0 = id1; c0l = c1l; c0h = c1h;
    id1 = id2; c1l = c2l; c1h = c2h;
  }
  // the groups' shares (loss: every lane of a group holds the group's term)
#pragma unroll
  for (int e = 0; e < EPL; e++) acc.c[e] = f64_across_groups<W>(acc.c[e]);
  if constexpr (W == 64 && MODE == 2) acc.c[0] = f64_group_sum<64>(acc.c[0]);   // (the terms sit one per quad there)
  return acc;
}

// EPL = coordinates per lane: 1 up to rank 64; 2 (W = 64: lane l holds the coordinates l and 64 + l) for ranks 65..128 -- round 5:
// the reference's default precision at the BASELINE ranks (R/model_WRMF.R:82) ran on the generic kernel's k^2 flops per
// non-zero until then, and WRMF kept such fits in fp32 behind a warning.
template <int W, bool IMPLICIT, int EPL = 1>
__global__ __launch_bounds__(256, EPL == 2 ? 2 : 1) void f64_cg_wave_kernel(F64Args a, int n_lo, int n_hi, int slot0) {   // rows of n_lo < n <= n_hi non-zeros
  static_assert(EPL == 1 || W == 64, "two coordinates per lane: the whole wave holds one vector");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* sG = reinterpret_cast<double*>(smem_raw);   // XtX (implicit), k x k
  __shared__ double sLoss[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int k = a.k;
  const int l = lane & (W - 1), g = lane / W;
  bool lk[EPL];
  int lc[EPL];
#pragma unroll
  for (int e = 0; e < EPL; e++) {
    lk[e] = l + 64 * e < k;
    lc[e] = min(l + 64 * e, k - 1);
  }
  // XtX in LDS: k x k up to rank 64; ranks 65..128 keep the lower triangle only (row i at i (i + 1) / 2: 66 KB instead of 128 KB at
  // rank 128, so that two workgroups fit a CU -- one wave per SIMD left the latency of every lane-crossing sum exposed)
  constexpr bool TRI = EPL > 1;
  int tl[EPL];
#pragma unroll
  for (int e = 0; e < EPL; e++) tl[e] = lc[e] * (lc[e] + 1) / 2;
  if (IMPLICIT) {
    for (int e = tid; e < k * k; e += 256) {
      if constexpr (TRI) {
        const int i = e / k, j = e - i * k;
        if (j <= i) sG[i * (i + 1) / 2 + j] = a.XtX[e];
      } else {
        sG[e] = a.XtX[e];
      }
    }
    __syncthreads();
  }
  using Vec = F64Vec<EPL>;
  // (G v)_l, v one coordinate per lane and register (replicated in the groups): v_m from lane m % 64, register m / 64
  auto gmv = [&](const Vec v) {
    Vec s0;
    int lo[EPL], hi[EPL];
#pragma unroll
    for (int e = 0; e < EPL; e++) {
      s0.c[e] = 0.0;
      lo[e] = __double2loint(v.c[e]);
      hi[e] = __double2hiint(v.c[e]);
    }
    // eight columns of G per trip, their LDS reads requested together: one column per trip -- read, wait, fma -- left a round trip
    // of the LDS exposed 128 times per product and four products per row (a third of the users' half at rank 128)
    auto column = [&](auto e2t, const int m, double (&gv)[EPL]) {
      constexpr int E2 = decltype(e2t)::value;
      const int mm = m + 64 * E2;
      const int tm = mm * (mm + 1) / 2;
#pragma unroll
      for (int e = 0; e < EPL; e++) {
        // XtX is symmetric; packed triangle: (row, col) with row >= col at row (row + 1) / 2 + col -- the lane's coordinate of
        // register e is below 64 for e = 0 and at least 64 for e = 1, so only e == E2 has to compare
        int at;
        if constexpr (!TRI) at = lc[e] + mm * k;
        else at = e == E2 ? (lc[e] >= mm ? tl[e] + mm : tm + lc[e]) : (e > E2 ? tl[e] + mm : tm + lc[e]);
        gv[e] = sG[at];
      }
    };
    static_for_f64<EPL>([&](auto e2t) {
      constexpr int E2 = decltype(e2t)::value;
      const int lim = min(64, k - 64 * E2);
      int m = 0;
      for (; m + 8 <= lim; m += 8) {
        double gv[8][EPL];
#pragma unroll
        for (int u = 0; u < 8; u++) column(e2t, m + u, gv[u]);
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const double vm = __hiloint2double(__builtin_amdgcn_readlane(hi[E2], m + u), __builtin_amdgcn_readlane(lo[E2], m + u));
#pragma unroll
          for (int e = 0; e < EPL; e++) s0.c[e] = fma(gv[u][e], vm, s0.c[e]);
        }
      }
      for (; m < lim; m++) {
        double gv[EPL];
        column(e2t, m, gv);
        const double vm = __hiloint2double(__builtin_amdgcn_readlane(hi[E2], m), __builtin_amdgcn_readlane(lo[E2], m));
#pragma unroll
        for (int e = 0; e < EPL; e++) s0.c[e] = fma(gv[e], vm, s0.c[e]);
      }
    });
#pragma unroll
    for (int e = 0; e < EPL; e++) s0.c[e] = lk[e] ? s0.c[e] : 0.0;
    return s0;
  };
  double wloss = 0.0;
  for (int row = blockIdx.x * 4 + wv; row < a.n_cols; row += gridDim.x * 4) {
    const int p1 = a.col_ptrs[row], n = a.col_ptrs[row + 1] - p1;
    double* yrow = a.Y + (size_t)row * k;
    if (!(n > n_lo && n <= n_hi)) continue;   // another launch's row
    if (n <= 0) {   // empty column -> zeros (wrmf_implicit.hpp:272-283, wrmf_explicit.hpp:133-144)
#pragma unroll
      for (int e = 0; e < EPL; e++)
        if (lane + 64 * e < k) yrow[lane + 64 * e] = 0.0;
      continue;
    }
    const double lam_use = IMPLICIT ? a.lambda : a.lambda * (a.dynamic_lambda ? (double)n : 1.0);
    auto pass = [&](const Vec v, auto mode_tag) {
      return f64_row_pass<W, IMPLICIT, EPL, decltype(mode_tag)::value>(a, k, p1, n, v, lk, lc, l, g, lane);
    };
    // sums over the coordinates (one group's lanes; the groups hold copies)
    auto dot = [&](const Vec u, const Vec v) {
      double sacc = 0.0;
#pragma unroll
      for (int e = 0; e < EPL; e++) sacc = fma(lk[e] ? u.c[e] : 0.0, v.c[e], sacc);
      return f64_group_sum<W>(sacc);
    };

    Vec x, r, pv;
#pragma unroll
    for (int e = 0; e < EPL; e++) x.c[e] = lk[e] ? yrow[l + 64 * e] : 0.0;   // warm start
    {
      const Vec t0 = pass(x, std::integral_constant<int, 0>{});
      Vec g0;
      if constexpr (IMPLICIT) g0 = gmv(x);
#pragma unroll
      for (int e = 0; e < EPL; e++) r.c[e] = t0.c[e] - (IMPLICIT ? g0.c[e] : lam_use * x.c[e]);
    }
    pv = r;
    double rsold = dot(r, r);
    for (int it = 0; it < a.cg_steps; it++) {
      Vec ap = pass(pv, std::integral_constant<int, 1>{});
      {
        Vec g1;
        if constexpr (IMPLICIT) g1 = gmv(pv);
#pragma unroll
        for (int e = 0; e < EPL; e++) ap.c[e] += IMPLICIT ? g1.c[e] : lam_use * pv.c[e];
      }
      const double alpha = rsold / dot(pv, ap);
#pragma unroll
      for (int e = 0; e < EPL; e++) {
        x.c[e] = fma(alpha, pv.c[e], x.c[e]);
        r.c[e] = fma(-alpha, ap.c[e], r.c[e]);
      }
      const double rsnew = dot(r, r);
      if (rsnew < kCgTolD) break;
      const double beta = rsnew / rsold;
#pragma unroll
      for (int e = 0; e < EPL; e++) pv.c[e] = fma(pv.c[e], beta, r.c[e]);
      rsold = rsnew;
    }
#pragma unroll
    for (int e = 0; e < EPL; e++)
      if (lane + 64 * e < k && (EPL > 1 || lane < k)) yrow[lane + 64 * e] = x.c[e];   // (group 0)
    const Vec lrow = pass(x, std::integral_constant<int, 2>{});
    wloss += lrow.c[0] + lam_use * dot(x, x);
  }
  if (lane == 0) sLoss[wv] = wloss;
  __syncthreads();
  if (tid == 0) a.loss_partials[slot0 + blockIdx.x] = (sLoss[0] + sLoss[1]) + (sLoss[2] + sLoss[3]);
}

bool f64_cg_wave_supported(const F64Args& a) {
  return a.solver == 1 && a.xb < 0 && !a.rhs_init && a.gbias == 0.0 && a.k1 == a.k && a.xoff == 0 && a.ioff == 0 && a.ooff == 0 &&
         !a.solve_empty && a.k >= 1 && a.k <= 128;
}

// ---- long rows ----
// The wave-per-row kernel finishes when its longest row does: the 1M x 100k matrix of the timing tools has an item with 154,282
// ratings, 48 k batches of sixteen gathers on ONE wave -- 58 ms (rank 64) / 80 ms (rank 128) of a launch whose other 99,999 rows take
// 12 (profiles/r05/r5p_*).  Rows beyond long_min non-zeros are therefore cut into chunks of chunk_len, listed when the matrix
// handle is made (wrmf_f64_capi.cpp: the column pointers are on the host there), and the five passes of their conjugate gradient
// run as launches of their own: every chunk a wave (f64_long_pass_kernel: the same f64_row_pass, partial sums to scratch), then one
// wave per long row adds the partials IN CHUNK ORDER and does the step's vector updates (f64_long_update_kernel).  Deterministic:
// no atomics, the order of every sum is fixed by the table.
// scratch: [n_chunks x k] partial sums | [n_long x k] r | [n_long x k] p | [n_long x 4] rsold, stopped, loss, -

template <int W, bool IMPLICIT, int EPL, int MODE>
__global__ __launch_bounds__(256, EPL == 2 ? 2 : 1) void f64_long_pass_kernel(F64Args a) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int k = a.k;
  const int l = lane & (W - 1), g = lane / W;
  const int c = blockIdx.x * 4 + wv;
  if (c >= a.n_chunks) return;
  bool lk[EPL];
  int lc[EPL];
#pragma unroll
  for (int e = 0; e < EPL; e++) {
    lk[e] = l + 64 * e < k;
    lc[e] = min(l + 64 * e, k - 1);
  }
  const int li = a.chunk_long[c], off = a.chunk_off[c], row = a.long_rows[li];
  const int p1 = a.col_ptrs[row], n = min(a.chunk_len, a.col_ptrs[row + 1] - p1 - off);
  double* part = a.long_scratch + (size_t)c * k;
  const double* pvec = a.long_scratch + ((size_t)a.n_chunks + a.n_long) * k + (size_t)li * k;
  const double* sc = a.long_scratch + ((size_t)a.n_chunks + 2 * (size_t)a.n_long) * k + 4 * (size_t)li;
  if (MODE == 1 && sc[1] != 0.0) return;   // the row's iteration has stopped (rsnew below the tolerance)
  const double* vsrc = MODE == 1 ? pvec : a.Y + (size_t)row * k;   // first residual and loss: at the row's own vector
  F64Vec<EPL> v;
#pragma unroll
  for (int e = 0; e < EPL; e++) v.c[e] = lk[e] ? vsrc[lc[e]] : 0.0;
  const F64Vec<EPL> acc = f64_row_pass<W, IMPLICIT, EPL, MODE>(a, k, p1 + off, n, v, lk, lc, l, g, lane);
  if constexpr (MODE == 2) {
    if (lane == 0) part[0] = acc.c[0];
  } else {
#pragma unroll
    for (int e = 0; e < EPL; e++)
      if (lane < W && lk[e]) part[l + 64 * e] = acc.c[e];
  }
}

// stage 0: after the first-residual pass; 1: after an operator pass (one conjugate-gradient step); 2: after the loss pass
template <bool IMPLICIT>
__global__ __launch_bounds__(256) void f64_long_update_kernel(F64Args a, int stage) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int k = a.k;
  const int li = blockIdx.x * 4 + wv;
  if (li >= a.n_long) return;
  bool lk[2];
  int lc[2];
#pragma unroll
  for (int e = 0; e < 2; e++) {
    lk[e] = lane + 64 * e < k;
    lc[e] = min(lane + 64 * e, k - 1);
  }
  using Vec = F64Vec<2>;
  const int row = a.long_rows[li];
  const int n = a.col_ptrs[row + 1] - a.col_ptrs[row];
  const double lam_use = IMPLICIT ? a.lambda : a.lambda * (a.dynamic_lambda ? (double)n : 1.0);
  double* yrow = a.Y + (size_t)row * k;
  double* rvec = a.long_scratch + (size_t)a.n_chunks * k + (size_t)li * k;
  double* pvec = rvec + (size_t)a.n_long * k;
  double* sc = a.long_scratch + ((size_t)a.n_chunks + 2 * (size_t)a.n_long) * k + 4 * (size_t)li;
  if (stage == 1 && sc[1] != 0.0) return;
  const int c0 = a.long_chunk0[li], c1 = a.long_chunk0[li + 1];
  auto load = [&](const double* src) {
    Vec v;
#pragma unroll
    for (int e = 0; e < 2; e++) v.c[e] = lk[e] ? src[lc[e]] : 0.0;
    return v;
  };
  auto store = [&](double* dst, const Vec v) {
#pragma unroll
    for (int e = 0; e < 2; e++)
      if (lk[e]) dst[lane + 64 * e] = v.c[e];
  };
  auto dot = [&](const Vec u, const Vec v) {
    double sacc = 0.0;
#pragma unroll
    for (int e = 0; e < 2; e++) sacc = fma(lk[e] ? u.c[e] : 0.0, v.c[e], sacc);
    return f64_group_sum<64>(sacc);
  };
  if (stage == 2) {
    double s = 0.0;
    for (int c = c0; c < c1; c++) s += a.long_scratch[(size_t)c * k];
    const Vec x = load(yrow);
    const double ls = s + lam_use * dot(x, x);
    if (lane == 0) sc[2] = ls;
    return;
  }
  Vec t;
  t.c[0] = t.c[1] = 0.0;
  for (int c = c0; c < c1; c++) {   // the chunks' shares, in chunk order
    const double* part = a.long_scratch + (size_t)c * k;
#pragma unroll
    for (int e = 0; e < 2; e++) t.c[e] += lk[e] ? part[lc[e]] : 0.0;
  }
  // (G v)_l, G = XtX (k x k, symmetric) from global memory: a few hundred rows per launch
  auto gmv = [&](const Vec v) {
    Vec s0;
    s0.c[0] = s0.c[1] = 0.0;
    const int lo[2] = {__double2loint(v.c[0]), __double2loint(v.c[1])}, hi[2] = {__double2hiint(v.c[0]), __double2hiint(v.c[1])};
#pragma unroll
    for (int e2 = 0; e2 < 2; e2++)
#pragma unroll 8
      for (int m = 0; m < min(64, k - 64 * e2); m++) {
        const double vm = __hiloint2double(__builtin_amdgcn_readlane(hi[e2], m), __builtin_amdgcn_readlane(lo[e2], m));
        const size_t col = (size_t)(m + 64 * e2) * k;
#pragma unroll
        for (int e = 0; e < 2; e++) s0.c[e] = fma(a.XtX[col + lc[e]], vm, s0.c[e]);
      }
#pragma unroll
    for (int e = 0; e < 2; e++) s0.c[e] = lk[e] ? s0.c[e] : 0.0;
    return s0;
  };
  Vec x = load(yrow);
  if (stage == 0) {
    Vec g0;
    if constexpr (IMPLICIT) g0 = gmv(x);
    Vec r;
#pragma unroll
    for (int e = 0; e < 2; e++) r.c[e] = t.c[e] - (IMPLICIT ? g0.c[e] : lam_use * x.c[e]);
    const double rsold = dot(r, r);
    store(rvec, r);
    store(pvec, r);
    if (lane == 0) {
      sc[0] = rsold;
      sc[1] = 0.0;
    }
    return;
  }
  Vec r = load(rvec), pv = load(pvec);
  const double rsold = sc[0];
  {
    Vec g1;
    if constexpr (IMPLICIT) g1 = gmv(pv);
#pragma unroll
    for (int e = 0; e < 2; e++) t.c[e] += IMPLICIT ? g1.c[e] : lam_use * pv.c[e];
  }
  const double alpha = rsold / dot(pv, t);
#pragma unroll
  for (int e = 0; e < 2; e++) {
    x.c[e] = fma(alpha, pv.c[e], x.c[e]);
    r.c[e] = fma(-alpha, t.c[e], r.c[e]);
  }
  const double rsnew = dot(r, r);
  store(yrow, x);
  store(rvec, r);
  if (rsnew < kCgTolD) {
    if (lane == 0) sc[1] = 1.0;
    return;
  }
  const double beta = rsnew / rsold;
#pragma unroll
  for (int e = 0; e < 2; e++) pv.c[e] = fma(pv.c[e], beta, r.c[e]);
  store(pvec, pv);
  if (lane == 0) sc[0] = rsnew;
}

// the long rows' loss terms, in row order, onto a slot the wave-per-row launch has already written
__global__ __launch_bounds__(64) void f64_long_loss_kernel(F64Args a, int slot) {
  if (threadIdx.x != 0) return;
  const double* sc = a.long_scratch + ((size_t)a.n_chunks + 2 * (size_t)a.n_long) * a.k;
  double s = 0.0;
  for (int li = 0; li < a.n_long; li++) s += sc[4 * (size_t)li + 2];
  a.loss_partials[slot] += s;
}

template <int W, bool IMPLICIT, int EPL>
hipError_t launch_f64_long_rows(const F64Args& a, int slots, hipStream_t s) {
  const dim3 gp((a.n_chunks + 3) / 4), gu((a.n_long + 3) / 4), b(256);
  hipLaunchKernelGGL((f64_long_pass_kernel<W, IMPLICIT, EPL, 0>), gp, b, 0, s, a);
  hipLaunchKernelGGL(f64_long_update_kernel<IMPLICIT>, gu, b, 0, s, a, 0);
  for (int it = 0; it < a.cg_steps; it++) {
    hipLaunchKernelGGL((f64_long_pass_kernel<W, IMPLICIT, EPL, 1>), gp, b, 0, s, a);
    hipLaunchKernelGGL(f64_long_update_kernel<IMPLICIT>, gu, b, 0, s, a, 1);
  }
  hipLaunchKernelGGL((f64_long_pass_kernel<W, IMPLICIT, EPL, 2>), gp, b, 0, s, a);
  hipLaunchKernelGGL(f64_long_update_kernel<IMPLICIT>, gu, b, 0, s, a, 2);
  hipLaunchKernelGGL(f64_long_loss_kernel, dim3(1), dim3(64), 0, s, a, slots - 1);
  return hipGetLastError();
}

// slots: the loss partials a.loss_partials[0 .. slots) are this call's (all written or zeroed)
// (Measured and not kept: the row's vectors resident in registers across the five passes -- one wave per row up to 256 non-zeros,
//  the four waves of a workgroup sharing a row up to 1024 -- lost to the streaming kernel on both sides of the 1M x 100k matrix:
//  200..256 registers and 64 unrolled steps per pass against 140 registers here; users 6.4 -> 12.5 ms, items 19 -> 23 ms.)
template <int W, int EPL = 1>
hipError_t launch_f64_cg_wave_w(const F64Args& a, int slots, hipStream_t s) {
  const size_t lds = !a.implicit ? 0 : (EPL > 1 ? (size_t)a.k * (a.k + 1) / 2 : (size_t)a.k * a.k) * sizeof(double);   // (rank 128: 66 KB)
  hipError_t err;
  const int grid = std::max(1, std::min((a.n_cols + 3) / 4, slots));
  const int n_hi = a.n_long > 0 ? a.long_min : 0x7fffffff;   // (longer rows: launch_f64_long_rows)
  if (grid < slots && (err = hipMemsetAsync(a.loss_partials + grid, 0, (size_t)(slots - grid) * sizeof(double), s)) != hipSuccess)
    return err;
  if (a.implicit) {
    auto kern = f64_cg_wave_kernel<W, true, EPL>;
    if (lds > 48 * 1024 &&
        (err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess)
      return err;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a, -1, n_hi, 0);
  } else {
    hipLaunchKernelGGL((f64_cg_wave_kernel<W, false, EPL>), dim3(grid), dim3(256), lds, s, a, -1, n_hi, 0);
  }
  if ((err = hipGetLastError()) != hipSuccess || a.n_long <= 0) return err;
  return a.implicit ? launch_f64_long_rows<W, true, EPL>(a, slots, s) : launch_f64_long_rows<W, false, EPL>(a, slots, s);
}

// ---- Gramian: partial[b] = sum over the block's columns of x x^T (lower-triangle tiles), then a fixed-order reduction ----
__global__ __launch_bounds__(256) void f64_gramian_partial_kernel(const double* __restrict__ X, int k, int64_t n, int KP,
                                                                  int CH, double* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* A = reinterpret_cast<double*>(smem_raw);
  const int LDA = KP + 1, tid = threadIdx.x;
  double* xs = A + (size_t)KP * LDA;
  const int T4 = KP / 4, ntiles = T4 * (T4 + 1) / 2;
  for (int e = tid; e < KP * LDA; e += 256) A[e] = 0.0;
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t e0 = (int64_t)blockIdx.x * per, e1 = min(n, e0 + per);
  __syncthreads();
  for (int64_t c0 = e0; c0 < e1; c0 += CH) {
    const int cn = (int)min((int64_t)CH, e1 - c0);
    for (int e = tid; e < cn * KP; e += 256) {
      const int j = e / KP, t = e - j * KP;
      xs[e] = t < k ? X[(size_t)(c0 + j) * k + t] : 0.0;
    }
    __syncthreads();
    rank_update_tiles<256>(A, LDA, xs, KP, nullptr, cn, ntiles);
    __syncthreads();
  }
  double* out = partial + (size_t)blockIdx.x * KP * LDA;
  for (int e = tid; e < KP * LDA; e += 256) out[e] = A[e];
}

__global__ __launch_bounds__(256) void f64_gramian_reduce_kernel(const double* __restrict__ partial, int blocks, int k, int KP,
                                                                 double ridge, double* __restrict__ XtX,
                                                                 double* __restrict__ sumsq) {
  const int LDA = KP + 1;
  const size_t mat = (size_t)KP * LDA;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < k * k) {
    const int c = e / k, i = e - c * k;
    const int lo = max(i, c), hi = min(i, c);   // the lower-triangle entry (lo, hi): the result is exactly symmetric
    double s = 0.0;
    for (int b = 0; b < blocks; b++) s += partial[(size_t)b * mat + lo + (size_t)hi * LDA];
    XtX[e] = i == c ? s + ridge : s;
  }
  if (blockIdx.x == 0 && sumsq) {   // trace before the ridge = sum(X^2): the regulariser term of the loss for free
    __shared__ double sdiag[128];
    if (threadIdx.x < 128) {
      double s = 0.0;
      if ((int)threadIdx.x < k)
        for (int b = 0; b < blocks; b++) s += partial[(size_t)b * mat + threadIdx.x + (size_t)threadIdx.x * LDA];
      sdiag[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double tr = 0.0;
      for (int i = 0; i < k; i++) tr += sdiag[i];
      sumsq[0] = tr;
    }
  }
}

constexpr int kRhsInitBlocksD = 256;
__global__ __launch_bounds__(128) void f64_rhs_init_partial_kernel(const double* __restrict__ X, int k, int off, int k1,
                                                                   int bias_row, double global_bias, int n,
                                                                   double* __restrict__ partial) {
  const int t = threadIdx.x;
  const int per = (n + gridDim.x - 1) / gridDim.x;
  const int e0 = blockIdx.x * per, e1 = min(n, e0 + per);
  double s = 0.0;
  if (t < k1)
    for (int e = e0; e < e1; e++)   // rhs_init = -X' (x_b + global_bias)  (wrmf_implicit.hpp:146-153; :110-112 without biases)
      s = fma(-X[(size_t)e * k + off + t], (bias_row >= 0 ? X[(size_t)e * k + bias_row] : 0.0) + global_bias, s);
  partial[(size_t)blockIdx.x * 128 + t] = s;
}
__global__ __launch_bounds__(128) void f64_rhs_init_reduce_kernel(const double* __restrict__ partial, int blocks,
                                                                  double* __restrict__ out) {
  const int t = threadIdx.x;
  double s = 0.0;
  for (int b = 0; b < blocks; b++) s += partial[(size_t)b * 128 + t];
  out[t] = s;
}

__global__ __launch_bounds__(256) void f64_weighted_sumsq_kernel(const double* __restrict__ X, int k, int64_t n,
                                                                 const double* __restrict__ w, double* __restrict__ partials) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * 4 + wv, n_waves = (int64_t)gridDim.x * 4;
  double s = 0.0;
  for (int64_t j = wave; j < n; j += n_waves) {
    double q = 0.0;
    for (int t = lane; t < k; t += 64) {
      const double v = X[(size_t)j * k + t];
      q = fma(v, v, q);
    }
    s += (w ? w[j] : 1.0) * q;
  }
  s = wave_sum_d(s);
  if (lane == 0) partials[wave] = s;
}

struct F64Geo {
  int KP, NT, CH, m2_in_lds;
  size_t lds;
};
constexpr size_t kF64LdsBudget = 156 * 1024;

F64Geo f64_geometry(int k1, int solver) {
  F64Geo g;
  g.KP = std::max(4, (k1 + 3) / 4 * 4);
  // (beyond rank 64 the matrix leaves room for one workgroup per CU: eight waves instead of four, two per SIMD, so that a wave's
  //  LDS and matrix-core latencies hide behind the other's -- nothing else would)
  g.NT = g.KP <= 32 ? 64 : g.KP <= 64 ? 256 : 512;
  const size_t mat = (size_t)g.KP * (g.KP + 1);
  const size_t vec = (size_t)9 * g.KP + 8;
  auto bytes = [&](int mats, int ch) { return (mats * mat + vec + (size_t)ch * (g.KP + 4)) * 8 + (size_t)(ch + 4) * 4 + 16; };
  g.m2_in_lds = (solver == 2 && bytes(2, 8) <= kF64LdsBudget) ? 1 : 0;
  const int mats = 1 + g.m2_in_lds;
  int ch = 64;
  while (ch > 4 && bytes(mats, ch) > kF64LdsBudget) ch >>= 1;
  // NNLS with the squared system in LDS (rank 33..64: 106 KB at 64 staged vectors): the sweeps of a row are ONE wave's serial
  // chain, and what fills the CU meanwhile is another workgroup -- a chunk of 16 leaves room for a second one (end of round 6)
  if (g.m2_in_lds) {
    int c = ch;
    while (c >= 8 && 2 * bytes(mats, c) > (size_t)158 * 1024) c >>= 1;
    if (c >= 8) ch = c;
  }
  // small systems: no need for the whole LDS (more workgroups per CU instead)
  if (g.KP <= 32) ch = std::min(ch, 32);
#ifdef RSP_AB   // (dev builds: the shipped library reads no environment variable)
  if (const char* e = std::getenv("RSPARSE_HIP_F64_CHUNK")) {   // dev: pin the chunk of staged vectors (occupancy experiments)
    const int c = std::atoi(e);
    if (c >= 4 && c <= 64 && (c & (c - 1)) == 0 && bytes(mats, c) <= kF64LdsBudget) ch = c;
  }
#endif
  g.CH = ch;
  g.lds = bytes(mats, ch);
  return g;
}

}  // namespace

// ---- explicit feedback with user/item biases and the conjugate-gradient solver: re-packed for the wave-per-row kernels ----
// The generic kernel takes the bias operands as offsets (xoff / xb / ioff / ooff) but pays k1^2 flops per non-zero and a workgroup
// per row; the wave-per-row conjugate-gradient kernels want plain matrices.  As in the fp32 layer (wrmf_bias.hip): X' = the k1
// kept rows of X, Y' = the warm start, r' = r - x_bias[index]; solve at rank k1; the result goes back to its rows of Y.
namespace {
__global__ __launch_bounds__(256) void f64_pack_rows_kernel(const double* __restrict__ src, int ld, int off, int k1, int64_t n,
                                                            double* __restrict__ dst) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < n * k1) {
    const int64_t j = e / k1;
    const int t = (int)(e - j * k1);
    dst[e] = src[j * ld + off + t];
  }
}
__global__ __launch_bounds__(256) void f64_unpack_rows_kernel(const double* __restrict__ src, int k1, int64_t n, int ld, int off,
                                                              double* __restrict__ dst) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < n * k1) {
    const int64_t j = e / k1;
    const int t = (int)(e - j * k1);
    dst[j * ld + off + t] = src[e];
  }
}
__global__ __launch_bounds__(256) void f64_shift_values_kernel(const double* __restrict__ vals, const int32_t* __restrict__ idx,
                                                               const double* __restrict__ X, int ld, int xb, int64_t nnz,
                                                               double* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < nnz) out[e] = vals[e] - X[(size_t)idx[e] * ld + xb];   // wrmf_explicit.hpp:89
}
}  // namespace
hipError_t launch_f64_pack_rows(const double* src, int ld, int off, int k1, int64_t n, double* dst, hipStream_t s) {
  if (n <= 0 || k1 <= 0) return hipSuccess;
  hipLaunchKernelGGL(f64_pack_rows_kernel, dim3((unsigned)((n * k1 + 255) / 256)), dim3(256), 0, s, src, ld, off, k1, n, dst);
  return hipGetLastError();
}
hipError_t launch_f64_unpack_rows(const double* src, int k1, int64_t n, int ld, int off, double* dst, hipStream_t s) {
  if (n <= 0 || k1 <= 0) return hipSuccess;
  hipLaunchKernelGGL(f64_unpack_rows_kernel, dim3((unsigned)((n * k1 + 255) / 256)), dim3(256), 0, s, src, k1, n, ld, off, dst);
  return hipGetLastError();
}
hipError_t launch_f64_shift_values(const double* vals, const int32_t* idx, const double* X, int ld, int xb, int64_t nnz,
                                   double* out, hipStream_t s) {
  if (nnz <= 0) return hipSuccess;
  hipLaunchKernelGGL(f64_shift_values_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, s, vals, idx, X, ld, xb, nnz, out);
  return hipGetLastError();
}

size_t f64_long_scratch_doubles(int k, int n_long, int n_chunks) { return ((size_t)n_chunks + 2 * (size_t)n_long) * k + 4 * (size_t)n_long; }
int f64_als_grid(int n_cols) { return std::max(1, std::min(n_cols, kF64MaxGrid)); }
bool f64_needs_m2_scratch(int k1, int solver) { return solver == 2 && !f64_geometry(k1, solver).m2_in_lds; }
size_t f64_m2_doubles_per_wg(int k1) {
  const int KP = std::max(4, (k1 + 3) / 4 * 4);
  return (size_t)KP * (KP + 1);
}

hipError_t launch_f64_als(const F64Args& a, hipStream_t s) {
  if (a.n_cols <= 0) return hipSuccess;
  const F64Geo g = f64_geometry(a.k1, a.solver);
  const int grid = f64_als_grid(a.n_cols);
  hipError_t err;
  if (f64_cg_wave_supported(a)) {   // the plain conjugate-gradient half-iteration: one wave per row, four rows per workgroup
    if (a.k <= 16) return launch_f64_cg_wave_w<16>(a, grid, s);
    if (a.k <= 32) return launch_f64_cg_wave_w<32>(a, grid, s);
    if (a.k <= 64) return launch_f64_cg_wave_w<64>(a, grid, s);
    return launch_f64_cg_wave_w<64, 2>(a, grid, s);   // ranks 65..128: two coordinates per lane (round 5)
  }
  if (g.NT == 64) {
    auto kern = f64_als_kernel<64>;
    if ((err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds)) != hipSuccess)
      return err;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), g.lds, s, a, g.KP, g.CH, g.m2_in_lds);
  } else if (g.NT == 256) {
    auto kern = f64_als_kernel<256>;
    if ((err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds)) != hipSuccess)
      return err;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), g.lds, s, a, g.KP, g.CH, g.m2_in_lds);
  } else {
    auto kern = f64_als_kernel<512>;
    if ((err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds)) != hipSuccess)
      return err;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), g.lds, s, a, g.KP, g.CH, g.m2_in_lds);
  }
  return hipGetLastError();
}

size_t f64_gramian_scratch_doubles(int k) {
  const int KP = std::max(4, (k + 3) / 4 * 4);
  return (size_t)kF64GramBlocks * KP * (KP + 1);
}

hipError_t launch_f64_gramian(const double* X, int k, int64_t n, double ridge, double* XtX, double* sumsq, double* scratch,
                              hipStream_t s) {
  const int KP = std::max(4, (k + 3) / 4 * 4);
  int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(kF64GramBlocks, (n + 63) / 64));
  const size_t mat = (size_t)KP * (KP + 1);
  int ch = 64;
  while (ch > 4 && (mat + (size_t)ch * KP) * 8 > kF64LdsBudget) ch >>= 1;
  const size_t lds = (mat + (size_t)ch * KP) * 8;
  auto kern = f64_gramian_partial_kernel;
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, s, X, k, n, KP, ch, scratch);
  if ((err = hipGetLastError()) != hipSuccess) return err;
  hipLaunchKernelGGL(f64_gramian_reduce_kernel, dim3((k * k + 255) / 256), dim3(256), 0, s, scratch, blocks, k, KP, ridge, XtX, sumsq);
  return hipGetLastError();
}

hipError_t launch_f64_rhs_init(const double* X, int k, int off, int k1, int bias_row, double global_bias, int n,
                               double* scratch, double* out, hipStream_t s) {
  hipLaunchKernelGGL(f64_rhs_init_partial_kernel, dim3(kRhsInitBlocksD), dim3(128), 0, s, X, k, off, k1, bias_row,
                     global_bias, n, scratch);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(f64_rhs_init_reduce_kernel, dim3(1), dim3(128), 0, s, scratch, kRhsInitBlocksD, out);
  return hipGetLastError();
}

hipError_t launch_f64_weighted_sumsq(const double* X, int k, int64_t n, const double* w, double* out, double* partials,
                                     hipStream_t s) {
  hipLaunchKernelGGL(f64_weighted_sumsq_kernel, dim3(256), dim3(256), 0, s, X, k, n, w, partials);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return launch_sum_partials(partials, 1024, out, s);
}

}  // namespace rsparse_hip
