// Ranks 129..256 (fp32; gfx950): the reference has no rank limit (arma::Mat<T>, inst/include/wrmf_implicit.hpp:103), the
// register- and tile-resident kernels of this library are built for rank <= 128.  One kernel family carries every variant of
// the half-iteration at the wider ranks -- implicit / explicit feedback, Cholesky (general-solver fallback included) /
// conjugate gradient / NNLS, the bias and global-bias operands of AlsArgs -- by assembling each row's k x k system in LDS as a
// PACKED lower triangle (k (k + 1) / 2 floats: 128.5 KB at rank 256), the scheme of wrmf_f64.hip with the storage halved:
//     assembly   4 x 4 register tiles of the lower triangle from chunks of the row staged in LDS
//     Cholesky   right-looking LL^T on the packed triangle, forward substitution riding along, backward by one wave; a
//                non-positive pivot: the system is assembled again, unpacked to a k x k scratch in global memory and solved by
//                Gaussian elimination with partial pivoting there (rare; arma::solve's fallback, wrmf_implicit.hpp:236)
//     CG         cg_solver_implicit / _global_bias / cg_solver_explicit (wrmf_implicit.hpp:8-57, wrmf_explicit.hpp:8-31) from
//                the warm start, A p from the assembled matrix; rsold / alpha / beta in double as the reference holds them
//     NNLS       c_nnls / scd_ls_update (nnls.hpp:10-48): XtX = lhs^T lhs (+ EPS) full in a global scratch, the sweeps by one
//                wave over the coordinates that can move
// It is the functional path of these ranks (one workgroup per row, k^2 flops per non-zero), not a tuned one; the bench line
// (rank 128) does not run through it.  Gramian of a factor matrix at these ranks: the same packed tiles (below).
#include <algorithm>
#include <cstdlib>

#include "wrmf_internal.h"
#include "wrmf_device.h"

namespace rsparse_hip {
namespace {

using namespace dev;

constexpr float kCgTolW = 1e-10f;
constexpr int kScdMaxIterW = 10000;
constexpr float kScdTolW = 1e-4f;
constexpr float kNnlsEpsW = 1e-16f;
constexpr int NT = 256;
#ifndef RSP_WIDE_ABL   // dev builds (timing only, results are garbage): bits switch phases of als_wide_kernel off
#define RSP_WIDE_ABL 0   // 1 trailing update, 2 rank-one updates, 4 loss pass, 8 Gramian copy, 16 backward substitution, 32 staging
#endif

__device__ __forceinline__ int tri(const int i) { return (i * (i + 1)) >> 1; }

__device__ __forceinline__ float wave_sum_f(float v) {   // butterfly: every lane ends with the same bits
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
// sum over the workgroup, every thread gets it; all threads must call it
__device__ __forceinline__ float block_sum_f(float v, float* red) {
  v = wave_sum_f(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ void tile_of_w(int t, int& ti, int& tj) {
  ti = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
  while ((ti + 1) * (ti + 2) / 2 <= t) ti++;
  while (ti * (ti + 1) / 2 > t) ti--;
  tj = t - ti * (ti + 1) / 2;
}
// packed A(i, c) += sum_j xs[j][i] w[j] xs[j][c] over the 4 x 4 tiles of the lower triangle
__device__ __forceinline__ void rank_update_packed(float* A, const float* xs, int kp, const float* w, int cn, int ntiles) {
  for (int t = threadIdx.x; t < ntiles; t += NT) {
    int ti, tj;
    tile_of_w(t, ti, tj);
    float acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int c = 0; c < 4; c++) acc[r][c] = 0.f;
    const float* xa = xs + 4 * ti;
    const float* xc = xs + 4 * tj;
    for (int j = 0; j < cn; j++) {
      const float wj = w ? w[j] : 1.f;
      float av[4], bv[4];
      // (xs is 16-byte aligned and kp a multiple of 4: one ds_read_b128 per operand instead of four 4-byte reads)
      const float4 a4 = *reinterpret_cast<const float4*>(xa + j * kp);
      const float4 c4 = *reinterpret_cast<const float4*>(xc + j * kp);
      av[0] = a4.x, av[1] = a4.y, av[2] = a4.z, av[3] = a4.w;
      bv[0] = wj * c4.x, bv[1] = wj * c4.y, bv[2] = wj * c4.z, bv[3] = wj * c4.w;
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) acc[r][c] = fmaf(av[r], bv[c], acc[r][c]);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int gi = 4 * ti + r, base = tri(gi);
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int gc = 4 * tj + c;
        if (gc <= gi) A[base + gc] += acc[r][c];
      }
    }
  }
}

struct WideArgs {
  const int32_t* col_ptrs;
  const int32_t* row_idx;
  const float* vals;
  const float* X;
  float* Y;
  const float* XtX;
  int n_cols, k;
  int implicit, solver, cg_steps, dynamic_lambda;
  double lambda_loss;
  const float* rhs_vals;   // see AlsArgs
  const float* loss_tgt;
  float loss_tgt_const;
  const float* rhs_init;
  float gbias;
  double* loss_partials;
  int* fail_counter;
  float* m2_scratch;   // NNLS: per workgroup k x (k + 1) floats
  float* lu_scratch;   // general solver: per workgroup k x k floats
  int n_lo;            // rows of at most this many non-zeros are another launch's (wrmf_wide_cg.hip); -1: none
};

__global__ __launch_bounds__(NT) void als_wide_kernel(WideArgs a, int KP, int CH) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sm = reinterpret_cast<float*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int k = a.k;
  const bool nnls = a.solver == 2, cg = a.solver == 1, implicit = a.implicit != 0;
  float* A = sm;                                   // packed lower triangle, KP (KP + 1) / 2
  float* rhs = A + (size_t)tri(KP);
  float* x = rhs + KP;
  float* r = x + KP;
  float* p = r + KP;
  float* ap = p + KP;
  float* sv = ap + KP;      // X_nnz (c - 1): the global-bias term of the first CG residual
  float* invd = sv + KP;    // Cholesky: 1 / L_jj
  float* red = invd + KP;   // 8
  float* xs = sm + (((red + 8) - sm + 3) & ~3);   // [CH][KP] staged factor vectors of the current chunk (16-byte aligned)
  float* cw = xs + (size_t)CH * KP;
  float* rw = cw + CH;
  float* lw = rw + CH;
  float* lt = lw + CH;
  int* sidx = reinterpret_cast<int*>(lt + CH);
  int* spiv = sidx + CH;
  const int LD2 = KP + 1;
  float* M2 = a.m2_scratch ? a.m2_scratch + (size_t)blockIdx.x * KP * LD2 : nullptr;
  float* F = a.lu_scratch ? a.lu_scratch + (size_t)blockIdx.x * k * k : nullptr;

  const int T4 = KP / 4, ntiles = T4 * (T4 + 1) / 2;
  int RT = 32;
  while (RT < k && RT < NT) RT <<= 1;
  const int CGR = NT / RT, ri = tid & (RT - 1), cgi = tid / RT;
  double wloss = 0.0;

  auto aget = [&](const int i, const int c) { return i >= c ? A[tri(i) + c] : A[tri(c) + i]; };
  auto stage = [&](const int p1, const int c0, const int cn) {
    if (tid < cn) {
      const int e = p1 + c0 + tid;
      const float c = a.vals[e];
      sidx[tid] = a.row_idx[e];
      cw[tid] = implicit ? c - 1.f : 1.f;
      rw[tid] = a.rhs_vals ? a.rhs_vals[e] : c;
      lw[tid] = implicit ? c : 1.f;
      lt[tid] = implicit ? (a.loss_tgt ? a.loss_tgt[e] : a.loss_tgt_const) : c;
    }
    __syncthreads();
    // thread t brings coordinate t of every vector, SG vectors in flight at a time (one load per trip with a division in front
    // of it was a round trip to HBM per KP / NT-th of a vector: 26 in a row for a chunk of 50 at order 132)
    constexpr int SG = 16;
    for (int t = tid; t < KP; t += NT) {
      for (int j0 = 0; j0 < cn; j0 += SG) {
        float v[SG];
#pragma unroll
        for (int u = 0; u < SG; u++) {
          const int j = min(j0 + u, cn - 1);
          v[u] = a.X[(size_t)sidx[j] * k + min(t, k - 1)];
        }
#pragma unroll
        for (int u = 0; u < SG; u++)
          if (j0 + u < cn) xs[(j0 + u) * KP + t] = t < k ? v[u] : 0.f;
      }
    }
    __syncthreads();
  };
  auto matrow = [&](const int t, const float* v) {   // (A v)[t]
    float s0 = 0.f, s1 = 0.f;
    const float* row = A + tri(t);
    int c = 0;
    for (; c + 1 <= t; c += 2) {
      s0 = fmaf(row[c], v[c], s0);
      s1 = fmaf(row[c + 1], v[c + 1], s1);
    }
    for (; c <= t; c++) s0 = fmaf(row[c], v[c], s0);
    for (c = t + 1; c < k; c++) s1 = fmaf(A[tri(c) + t], v[c], s1);
    return s0 + s1;
  };

  for (int row = blockIdx.x; row < a.n_cols; row += gridDim.x) {
    const int p1 = a.col_ptrs[row], n = a.col_ptrs[row + 1] - p1;
    float* yrow = a.Y + (size_t)row * k;
    if (n <= a.n_lo) continue;
    if (n <= 0 && !a.rhs_init) {   // empty column -> zeros (wrmf_implicit.hpp:272-283, wrmf_explicit.hpp:133-144)
      for (int t = tid; t < k; t += NT) yrow[t] = 0.f;
      continue;
    }
    const float lam_use = implicit ? 0.f : (float)(a.lambda_loss * (a.dynamic_lambda ? (double)(float)n : 1.0));
    auto assemble = [&]() {
      __syncthreads();
      for (int e = tid; e < tri(KP); e += NT) A[e] = 0.f;
      __syncthreads();
      if (implicit && !(RSP_WIDE_ABL & 8)) {
        // thread i brings row i of the lower triangle, GU columns in flight at a time (coalesced over i; was one load per trip
        // behind a division: k k / NT = 68 round trips to L2 in a row at order 132)
        constexpr int GU = 16;
        for (int i = tid; i < k; i += NT) {
          float* rowi = A + tri(i);
          for (int c0 = 0; c0 <= i; c0 += GU) {
            float v[GU];
#pragma unroll
            for (int u = 0; u < GU; u++) v[u] = a.XtX[i + (size_t)min(c0 + u, i) * k];
#pragma unroll
            for (int u = 0; u < GU; u++)
              if (c0 + u <= i) rowi[c0 + u] = v[u];
          }
        }
      }
      for (int t = tid; t < KP; t += NT) {
        rhs[t] = (a.rhs_init && t < k) ? a.rhs_init[t] : 0.f;
        sv[t] = 0.f;
      }
      __syncthreads();
      for (int c0 = 0; c0 < n; c0 += CH) {
        const int cn = min(CH, n - c0);
        if (!(RSP_WIDE_ABL & 32)) stage(p1, c0, cn);
        if (!(RSP_WIDE_ABL & 2)) rank_update_packed(A, xs, KP, cw, cn, ntiles);
        for (int t = tid; t < k; t += NT) {
          float s1 = 0.f, s2 = 0.f;
          for (int j = 0; j < cn; j++) {
            const float xv = xs[j * KP + t];
            s1 = fmaf(rw[j], xv, s1);
            s2 = fmaf(cw[j], xv, s2);
          }
          rhs[t] += s1;
          sv[t] += s2;
        }
        __syncthreads();
      }
      if (!implicit)
        for (int t = tid; t < k; t += NT) A[tri(t) + t] += lam_use;
      __syncthreads();
    };
    assemble();
    for (int t = tid; t < KP; t += NT) x[t] = t < k ? yrow[t] : 0.f;   // warm start (CG, NNLS)
    __syncthreads();

    if (cg) {
      float part = 0.f;
      for (int t = tid; t < k; t += NT) {
        const float rr = (rhs[t] - matrow(t, x)) - a.gbias * sv[t];
        r[t] = rr;
        p[t] = rr;
        part = fmaf(rr, rr, part);
      }
      float rsold = block_sum_f(part, red);
      for (int it = 0; it < a.cg_steps; it++) {
        __syncthreads();
        part = 0.f;
        for (int t = tid; t < k; t += NT) {
          const float s = matrow(t, p);
          ap[t] = s;
          part = fmaf(p[t], s, part);
        }
        const float pap = block_sum_f(part, red);
        const float alpha = (float)((double)rsold / (double)pap);   // double scalars (wrmf_implicit.hpp:18)
        part = 0.f;
        for (int t = tid; t < k; t += NT) {
          x[t] = fmaf(alpha, p[t], x[t]);
          const float rr = fmaf(-alpha, ap[t], r[t]);
          r[t] = rr;
          part = fmaf(rr, rr, part);
        }
        const float rsnew = block_sum_f(part, red);
        if (rsnew < kCgTolW) break;
        const float beta = (float)((double)rsnew / (double)rsold);
        for (int t = tid; t < k; t += NT) p[t] = fmaf(p[t], beta, r[t]);
        rsold = rsnew;
      }
      __syncthreads();
    } else if (nnls) {
      // XtX = lhs^T lhs + EPS I (full, global scratch), mu = XtX init - lhs^T rhs
      for (int t = tid; t < ntiles; t += NT) {
        int ti, tj;
        tile_of_w(t, ti, tj);
        float acc[4][4];
#pragma unroll
        for (int rr = 0; rr < 4; rr++)
#pragma unroll
          for (int c = 0; c < 4; c++) acc[rr][c] = 0.f;
        for (int m = 0; m < k; m++) {
          float av[4], bv[4];
#pragma unroll
          for (int rr = 0; rr < 4; rr++) av[rr] = aget(4 * ti + rr, m);
#pragma unroll
          for (int c = 0; c < 4; c++) bv[c] = aget(4 * tj + c, m);
#pragma unroll
          for (int rr = 0; rr < 4; rr++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc[rr][c] = fmaf(av[rr], bv[c], acc[rr][c]);
        }
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            const int gi = 4 * ti + rr, gc = 4 * tj + c;
            const float v = acc[rr][c] + (gi == gc ? kNnlsEpsW : 0.f);
            M2[gi + (size_t)gc * LD2] = v;
            M2[gc + (size_t)gi * LD2] = v;
          }
      }
      __syncthreads();
      for (int t = tid; t < k; t += NT) {
        float s0 = 0.f;
        for (int c = 0; c < k; c++) s0 = fmaf(M2[t + (size_t)c * LD2], x[c], s0);
        p[t] = s0 - matrow(t, rhs);
      }
      __syncthreads();
      if (wv == 0) {   // scd_ls_update: lane l owns coordinates l + 64 q
        float h[4], mu[4], dg[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int c = lane + 64 * q;
          h[q] = c < k ? x[c] : 0.f;
          mu[q] = c < k ? p[c] : 0.f;
          dg[q] = c < k ? M2[c + (size_t)c * LD2] : 1.f;
        }
        for (int t = 0; t < kScdMaxIterW; t++) {
          float rel = 0.f;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int lim = min(64, k - 64 * q);
            if (lim <= 0) break;
            const unsigned long long in_range = lim >= 64 ? ~0ull : ((1ull << lim) - 1ull);
            unsigned long long act = __ballot(!(h[q] == 0.f && mu[q] >= 0.f)) & in_range;   // idle visits change nothing (nnls.hpp:24)
            while (act) {
              const int l = __builtin_ctzll(act);
              const int c = 64 * q + l;
              const float old_v = readlane_f(h[q], l);
              float new_v = old_v - readlane_f(mu[q], l) / readlane_f(dg[q], l);
              if (new_v < 0.f) new_v = 0.f;
              const float diff = new_v - old_v;
              const unsigned long long above = l >= 63 ? 0ull : (~0ull << (l + 1));
              if (diff != 0.f) {
                if (lane == l) h[q] = new_v;
                const float* col = M2 + (size_t)c * LD2;
#pragma unroll
                for (int q2 = 0; q2 < 4; q2++)
                  if (lane + 64 * q2 < k) mu[q2] = fmaf(diff, col[lane + 64 * q2], mu[q2]);
                rel = fmaxf(rel, fabsf(diff) / (fabsf(old_v) + kNnlsEpsW));
                act = __ballot(!(h[q] == 0.f && mu[q] >= 0.f)) & in_range & above;
              } else {
                act &= above;
              }
            }
          }
          if (rel <= kScdTolW) break;
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (lane + 64 * q < k) x[lane + 64 * q] = h[q];
      }
      __syncthreads();
    } else {
      // ---- Cholesky on the packed triangle; z = L^-1 rhs rides along in x ----
      bool ok = true;
      // Two pivots per pass over the trailing matrix: column j is scaled, column j + 1 (and the right-hand side) takes ITS update
      // alone, is scaled in turn, and the rest of the matrix takes both updates in one read-modify-write -- element by element
      // the same two multiply-adds in the same order, two thirds of the LDS traffic.  Column j lives in p, column j + 1 in ap
      // (both free in this branch) as contiguous vectors: the update reads them by column index instead of walking the packed
      // triangle with a stride of c + 1 floats.  Rows over 32 threads, the columns of a row over 8 (a thread per ROW -- what
      // RT = NT gives at these orders -- left the longest row to one thread, pivot after pivot); eight elements per trip, all
      // reads before the first write: the compiler cannot know that the elements differ, and one read-multiply-write per trip
      // was an LDS round trip per multiply-add.
      constexpr int RTC = 32, CGC = NT / RTC, CU = 8;
      const int rc = tid & (RTC - 1), cc = tid / RTC;
      for (int j = 0; j < k; j += 2) {
        const int j1 = j + 1;
        __syncthreads();
        const float d = A[tri(j) + j];
        if (!(d > 0.f)) {
          ok = false;
          break;
        }
        const float dinv = 1.f / sqrtf(d);
        for (int i = j + 1 + tid; i < k; i += NT) {
          const float l = A[tri(i) + j] * dinv;
          A[tri(i) + j] = l;
          p[i] = l;
        }
        if (tid == 0) {
          invd[j] = dinv;
          x[j] = rhs[j] * dinv;
        }
        __syncthreads();
        if (j1 >= k) break;
        const float zj = x[j], lj1 = p[j1];
        for (int i = j1 + tid; i < k; i += NT) {   // column j + 1 and the right-hand side: pivot j's update
          const float li = p[i];
          A[tri(i) + j1] = fmaf(-li, lj1, A[tri(i) + j1]);
          rhs[i] = fmaf(-li, zj, rhs[i]);
        }
        __syncthreads();
        const float d1 = A[tri(j1) + j1];
        if (!(d1 > 0.f)) {
          ok = false;
          break;
        }
        const float dinv1 = 1.f / sqrtf(d1);
        for (int i = j1 + 1 + tid; i < k; i += NT) {
          const float l = A[tri(i) + j1] * dinv1;
          A[tri(i) + j1] = l;
          ap[i] = l;
        }
        if (tid == 0) {
          invd[j1] = dinv1;
          x[j1] = rhs[j1] * dinv1;
        }
        __syncthreads();
        const float zj1 = x[j1];
        for (int i = j + 2 + rc; i < k && !(RSP_WIDE_ABL & 1); i += RTC) {
          float* rowi = A + tri(i);
          const float li = p[i], li1 = ap[i];
          int c = j + 2 + cc;
          for (; c + (CU - 1) * CGC <= i; c += CU * CGC) {
            float av[CU], bv[CU], b1[CU];
#pragma unroll
            for (int u = 0; u < CU; u++) {
              av[u] = rowi[c + u * CGC];
              bv[u] = p[c + u * CGC];
              b1[u] = ap[c + u * CGC];
            }
#pragma unroll
            for (int u = 0; u < CU; u++) rowi[c + u * CGC] = fmaf(-li1, b1[u], fmaf(-li, bv[u], av[u]));
          }
          for (; c <= i; c += CGC) rowi[c] = fmaf(-li1, ap[c], fmaf(-li, p[c], rowi[c]));
          if (cc == 0) rhs[i] = fmaf(-li1, zj1, rhs[i]);
        }
      }
      __syncthreads();
      if (ok) {
        if (wv == 0 && !(RSP_WIDE_ABL & 16)) {   // L^T y = z: lane l holds entries l + 64 q; row m of L is contiguous in the packed triangle
          float z[4];
#pragma unroll
          for (int q = 0; q < 4; q++) z[q] = lane + 64 * q < k ? x[lane + 64 * q] : 0.f;
          for (int m = k - 1; m >= 0; m--) {
            const int src = m & 63, qm = m >> 6;
            const float zsel = qm == 0 ? z[0] : (qm == 1 ? z[1] : (qm == 2 ? z[2] : z[3]));
            const float ym = __shfl(zsel, src) * invd[m];
            const float* rowm = A + tri(m);
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const int i = lane + 64 * q;
              if (i == m) z[q] = ym;
              else if (i < m) z[q] = fmaf(-rowm[i], ym, z[q]);
            }
          }
#pragma unroll
          for (int q = 0; q < 4; q++)
            if (lane + 64 * q < k) x[lane + 64 * q] = z[q];
        }
        __syncthreads();
      } else {
        // ---- the general solver on the system assembled again, unpacked into global memory ----
        assemble();
        for (int e = tid; e < k * k; e += NT) {
          const int c = e / k, i = e - c * k;
          F[e] = aget(i, c);
        }
        bool singular = false;
        for (int c = 0; c < k; c++) {
          __syncthreads();
          if (wv == 0) {
            float best = -1.f;
            int bi = c;
            for (int i = c + lane; i < k; i += 64) {
              const float v = fabsf(F[i + (size_t)c * k]);
              if (v > best) { best = v; bi = i; }
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
              const float ov = __shfl_xor(best, m);
              const int oi = __shfl_xor(bi, m);
              if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            if (lane == 0) {
              spiv[0] = bi;
              spiv[1] = best > 0.f ? 0 : 1;
            }
          }
          __syncthreads();
          if (spiv[1]) {
            singular = true;
            break;
          }
          const int piv = spiv[0];
          if (piv != c) {
            for (int m = tid; m < k; m += NT) {
              const float t0 = F[c + (size_t)m * k];
              F[c + (size_t)m * k] = F[piv + (size_t)m * k];
              F[piv + (size_t)m * k] = t0;
            }
            if (tid == 0) {
              const float t0 = rhs[c];
              rhs[c] = rhs[piv];
              rhs[piv] = t0;
            }
          }
          __syncthreads();
          const float pinv = 1.f / F[c + (size_t)c * k];
          const float bc = rhs[c];
          for (int i = c + 1 + ri; i < k; i += RT) {
            const float f = F[i + (size_t)c * k] * pinv;
            if (f != 0.f) {
              for (int m = c + 1 + cgi; m < k; m += CGR) F[i + (size_t)m * k] = fmaf(-f, F[c + (size_t)m * k], F[i + (size_t)m * k]);
              if (cgi == 0) rhs[i] = fmaf(-f, bc, rhs[i]);
            }
          }
        }
        __syncthreads();
        if (!singular) {
          if (tid == 0) {   // U y = b (serial: this path is rare)
            for (int m = k - 1; m >= 0; m--) {
              float v = rhs[m];
              for (int c = m + 1; c < k; c++) v = fmaf(-F[m + (size_t)c * k], x[c], v);
              x[m] = v / F[m + (size_t)m * k];
            }
          }
        } else {
          for (int t = tid; t < k; t += NT) x[t] = 0.f;
        }
        if (tid == 0 && a.fail_counter) {
          atomicAdd(a.fail_counter + 2, 1);
          if (singular) atomicAdd(a.fail_counter + 3, 1);
        }
        __syncthreads();
      }
    }

    // ---- write back, loss term ----
    for (int t = tid; t < k; t += NT) yrow[t] = x[t];
    float lpart = 0.f;
    for (int c0 = 0; c0 < n && !(RSP_WIDE_ABL & 4); c0 += CH) {
      const int cn = min(CH, n - c0);
      stage(p1, c0, cn);
      for (int j = wv; j < cn; j += 4) {
        float s = 0.f;
        for (int t = lane; t < k; t += 64) s = fmaf(xs[j * KP + t], x[t], s);
        s = wave_sum_f(s);
        const float dlt = lt[j] - s;
        lpart = fmaf(lw[j] * dlt, dlt, lpart);
      }
      __syncthreads();
    }
    float yy = 0.f;
    for (int t = tid; t < k; t += NT) yy = fmaf(x[t], x[t], yy);
    const float lsum = block_sum_f(lane == 0 ? lpart : 0.f, red);
    const float ysum = block_sum_f(yy, red);
    if (tid == 0) wloss += implicit ? (double)lsum + a.lambda_loss * (double)ysum : (double)(lsum + lam_use * ysum);
  }
  if (tid == 0) a.loss_partials[blockIdx.x] = wloss;
}

// ---- Gramian at these ranks: partial[b] = packed sum over the block's columns of x x^T, then a fixed-order reduction ----
__global__ __launch_bounds__(NT) void wide_gramian_partial_kernel(const float* __restrict__ X, int k, int64_t n, int KP, int CH,
                                                                  float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* A = reinterpret_cast<float*>(smem_raw);
  float* xs = A + (((size_t)tri(KP) + 3) & ~(size_t)3);   // (16-byte aligned: rank_update_packed reads it in 16-byte pieces)
  const int tid = threadIdx.x;
  const int T4 = KP / 4, ntiles = T4 * (T4 + 1) / 2;
  for (int e = tid; e < tri(KP); e += NT) A[e] = 0.f;
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t e0 = (int64_t)blockIdx.x * per, e1 = min(n, e0 + per);
  __syncthreads();
  for (int64_t c0 = e0; c0 < e1; c0 += CH) {
    const int cn = (int)min((int64_t)CH, e1 - c0);
    constexpr int SG = 16;   // rows in flight per thread (thread t brings coordinate t; see als_wide_kernel's stage)
    for (int t = tid; t < KP; t += NT) {
      for (int j0 = 0; j0 < cn; j0 += SG) {
        float v[SG];
#pragma unroll
        for (int u = 0; u < SG; u++) v[u] = X[(size_t)(c0 + min(j0 + u, cn - 1)) * k + min(t, k - 1)];
#pragma unroll
        for (int u = 0; u < SG; u++)
          if (j0 + u < cn) xs[(j0 + u) * KP + t] = t < k ? v[u] : 0.f;
      }
    }
    __syncthreads();
    rank_update_packed(A, xs, KP, nullptr, cn, ntiles);
    __syncthreads();
  }
  float* out = partial + (size_t)blockIdx.x * tri(KP);
  for (int e = tid; e < tri(KP); e += NT) out[e] = A[e];
}

__global__ __launch_bounds__(NT) void wide_gramian_reduce_kernel(const float* __restrict__ partial, int blocks, int k, int KP,
                                                                 float ridge, float* __restrict__ XtX,
                                                                 double* __restrict__ sumsq) {
  const size_t mat = (size_t)tri(KP);
  const int e = blockIdx.x * NT + threadIdx.x;
  if (e < k * k) {
    const int c = e / k, i = e - c * k;
    const int lo = max(i, c), hi = min(i, c);
    float s = 0.f;
    for (int b = 0; b < blocks; b++) s += partial[(size_t)b * mat + tri(lo) + hi];
    XtX[e] = i == c ? s + ridge : s;
  }
  if (blockIdx.x == 0 && sumsq) {   // trace before the ridge = sum(X^2)
    __shared__ double sdiag[256];
    double s = 0.0;
    if ((int)threadIdx.x < k) {
      float f = 0.f;
      for (int b = 0; b < blocks; b++) f += partial[(size_t)b * mat + tri(threadIdx.x) + threadIdx.x];
      s = (double)f;
    }
    sdiag[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      double tr = 0.0;
      for (int i = 0; i < k; i++) tr += sdiag[i];
      sumsq[0] = tr;
    }
  }
}

constexpr size_t kWideLds = 156 * 1024;
constexpr int kWideGramBlocks = 256;
int wide_kp(int k) { return (k + 3) / 4 * 4; }

}  // namespace

bool wide_supported(int k) { return k > 128 && k <= 256; }
int wide_als_grid(int n_cols) { return std::max(1, std::min(n_cols, 256 * 4)); }
size_t wide_m2_floats_per_wg(int k) { return (size_t)wide_kp(k) * (wide_kp(k) + 1); }
size_t wide_gramian_scratch_floats(int k) { return (size_t)kWideGramBlocks * (wide_kp(k) * (wide_kp(k) + 1) / 2); }

// the half-iteration of `a` (system order a.k in 129..256) on `grid` = wide_als_grid(n_cols) workgroups; loss partials
// a.loss_partials[0 .. grid)
hipError_t launch_als_wide(const AlsArgs& a, bool implicit, unsigned solver, float* m2_scratch, float* lu_scratch,
                           hipStream_t s, int n_lo) {
  if (a.n_cols <= 0) return hipSuccess;
  WideArgs w;
  w.n_lo = n_lo;
  w.col_ptrs = a.col_ptrs; w.row_idx = a.row_idx; w.vals = a.vals; w.X = a.X; w.Y = a.Y; w.XtX = a.XtX;
  w.n_cols = a.n_cols; w.k = a.k; w.implicit = implicit ? 1 : 0; w.solver = (int)solver; w.cg_steps = a.cg_steps;
  w.dynamic_lambda = a.dynamic_lambda; w.lambda_loss = a.lambda_loss;
  w.rhs_vals = a.rhs_vals; w.loss_tgt = a.loss_tgt; w.loss_tgt_const = a.loss_tgt_const; w.rhs_init = a.rhs_init;
  w.gbias = a.gbias;
  w.loss_partials = a.loss_partials; w.fail_counter = a.fail_counter;
  w.m2_scratch = solver == 2 ? m2_scratch : nullptr;
  w.lu_scratch = lu_scratch;
  const int KP = wide_kp(a.k);
  const size_t fixed = ((size_t)KP * (KP + 1) / 2 + 7 * (size_t)KP + 8) * 4;
  auto lds_of = [&](int c) { return fixed + (size_t)c * (KP + 4) * 4 + (size_t)(c + 4) * 4 + 16; };
  int ch = 64;
  while (ch > 4 && lds_of(ch) > kWideLds) ch >>= 1;
  // A row is a chain of barriers and LDS round trips: what hides one workgroup's is ANOTHER workgroup on the CU.  Where a smaller
  // chunk of staged vectors (16 at least: one trip of the staging loop) buys a SECOND resident workgroup, take it: order 160,
  // 98 KB -> 77 KB, 335 -> 194 ms per iteration.  (A third one -- order 132 at 16 vectors per chunk -- measured 13 % slower than
  // two at 64: profiles/r06/r6wide_ab_*.  RSPARSE_HIP_WIDE_CHUNK pins the chunk for such measurements in -DRSP_AB builds.)
  if (2 * lds_of(ch) > (size_t)158 * 1024) {
    int c = ch;
    while (c >= 16 && 2 * lds_of(c) > (size_t)158 * 1024) c >>= 1;
    if (c >= 16) ch = c;
  }
#ifdef RSP_AB   // (dev builds: the shipped library reads no environment variable)
  if (const char* e = std::getenv("RSPARSE_HIP_WIDE_CHUNK")) {
    const int c = std::atoi(e);
    if (c >= 4 && c <= 64 && (c & (c - 1)) == 0 && lds_of(c) <= kWideLds) ch = c;
  }
#endif
  const size_t lds = lds_of(ch);
  auto kern = als_wide_kernel;
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL(kern, dim3(wide_als_grid(a.n_cols)), dim3(NT), lds, s, w, KP, ch);
  return hipGetLastError();
}

hipError_t launch_gramian_wide(const float* X, int k, int64_t n, float ridge, float* XtX, double* sumsq, float* scratch,
                               hipStream_t s) {
  const int KP = wide_kp(k);
  const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(kWideGramBlocks, (n + 63) / 64));
  const size_t mat = (size_t)KP * (KP + 1) / 2;
  int ch = 64;
  while (ch > 4 && (mat + (size_t)ch * KP) * 4 + 16 > kWideLds) ch >>= 1;
  const size_t lds = (mat + (size_t)ch * KP) * 4 + 16;   // (+ the alignment of the staged rows)
  auto kern = wide_gramian_partial_kernel;
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT), lds, s, X, k, n, KP, ch, scratch);
  if ((err = hipGetLastError()) != hipSuccess) return err;
  hipLaunchKernelGGL(wide_gramian_reduce_kernel, dim3((k * k + NT - 1) / NT), dim3(NT), 0, s, scratch, blocks, k, KP, ridge, XtX,
                     sumsq);
  return hipGetLastError();
}

}  // namespace rsparse_hip
