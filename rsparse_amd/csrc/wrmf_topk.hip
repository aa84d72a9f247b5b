// top-k of the dense product  scores = U * V^T  per user, fused (gfx950, wave64).
//
// Replaces top_product() (src/matrix_top_product.cpp:20-102), the C++ behind find_top_product()
// (R/utils.R:31-59) and `$predict` (R/MatrixFactorizationRecommender.R:24-78): for every user row j,
// scores_i = u_j . v_i over all items, skip the items of the user's `not_recommend` row (sorted CSR) and the
// globally excluded items, keep the k best in a min-heap (`q.top().first < val` -> on ties the earlier item
// stays), emit them best first -- equal scores come out with the larger index first, because the heap pops
// (score, index) pairs in ascending pair order and the output is filled from the end (:88-95).
//
// One 256-thread workgroup per block of 32 users.  The user block is the MFMA A operand and stays in
// registers (k/2 VGPRs); each of the 4 waves walks its own 32-item tiles: the tile is loaded with coalesced
// 16-byte reads, staged in LDS with an odd row stride, read back as B fragments, and multiplied with
// v_mfma_f32_32x32x2_f32 (exact fp32).  A score survives only if it beats the user's current k-th best
// (threshold in LDS); survivors that pass the exclusion checks (binary searches) are appended to the user's
// LDS buffer; once per round of 4 tiles, buffers holding more than k entries are reduced to their top k by
// rank counting and the threshold is raised.  After the first few tiles almost nothing survives.
#include "wrmf_internal.h"
#include "wrmf_device.h"

namespace rsparse_hip {
namespace {

using namespace dev;

typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr int kTopUsers = 32;    // users per workgroup
constexpr int kTopCap = 256;     // candidate buffer per user (k <= kTopCap - 128)
constexpr int kTopWaves = 4;

template <int KP>
struct TopSmem {
  static constexpr int LDT = KP + 1;  // odd stride: conflict-free column reads (ds_read_b32)
  static constexpr size_t tile_floats = (size_t)kTopWaves * 32 * LDT;
  static constexpr size_t buf_floats = (size_t)kTopUsers * kTopCap;           // values
  static constexpr size_t bytes = (tile_floats + 2 * buf_floats + 8 * kTopUsers) * 4 + 64 +
                                  (size_t)kTopWaves * kTopCap * 8;            // compaction scratch
};

// is `item` in the sorted list a[0..n) ?
__device__ __forceinline__ bool sorted_contains(const int32_t* __restrict__ a, int n, int item) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const int v = a[mid];
    if (v < item) lo = mid + 1;
    else hi = mid;
  }
  return lo < n && a[lo] == item;
}

// (va, ia) ranks before (vb, ib) when selecting: larger score first, earlier item on ties
__device__ __forceinline__ bool sel_before(float va, int ia, float vb, int ib) {
  return va > vb || (va == vb && ia < ib);
}

template <int KP>
__global__ __launch_bounds__(256) void top_product_kernel(const float* __restrict__ U, const float* __restrict__ V,
                                                          int n_users, int n_items, int k_rank, int topk,
                                                          const int32_t* __restrict__ nr_ptr,
                                                          const int32_t* __restrict__ nr_idx,
                                                          const int32_t* __restrict__ excl, int n_excl,
                                                          float glob_mean, int32_t* __restrict__ res,
                                                          float* __restrict__ scores_out) {
  using SM = TopSmem<KP>;
  constexpr int LDT = SM::LDT, NK2 = KP / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sTile = reinterpret_cast<float*>(smem);
  float* sVal = sTile + SM::tile_floats;                            // [32][CAP]
  int* sIdx = reinterpret_cast<int*>(sVal + SM::buf_floats);        // [32][CAP]
  int* sCnt = sIdx + SM::buf_floats;                                // [32]
  float* sThr = reinterpret_cast<float*>(sCnt + kTopUsers);         // [32]
  int* sNeed = reinterpret_cast<int*>(sThr + kTopUsers);            // [32] spare / flags
  float* sTmpV = reinterpret_cast<float*>(sNeed + 2 * kTopUsers + 4);  // [4][CAP]
  int* sTmpI = reinterpret_cast<int*>(sTmpV + kTopWaves * kTopCap);    // [4][CAP]

  const int tid = threadIdx.x, lane = tid & 63, wv = rfl(tid >> 6);
  const int u0 = blockIdx.x * kTopUsers;
  const int col = lane & 31, half = lane >> 5;
  if (tid < kTopUsers) {
    sCnt[tid] = 0;
    sThr[tid] = -INFINITY;
    sNeed[tid] = 0;   // entries [0, sNeed) of the user's buffer are the heap as the last reduction left it
  }
  // A operand: lane holds U[u0 + (lane & 31)][2t + half], t = 0..KP/2-1 (zero beyond the matrix)
  float afrag[NK2];
  {
    const int u = u0 + col;
#pragma unroll
    for (int t = 0; t < NK2; t++) {
      const int kk = 2 * t + half;
      afrag[t] = (u < n_users && kk < k_rank) ? U[(size_t)u * k_rank + kk] : 0.f;
    }
  }
  float* tile = sTile + wv * 32 * LDT;
  for (int e = lane; e < 32 * LDT; e += 64) tile[e] = 0.f;
  __syncthreads();

  const int n_tiles = (n_items + 31) / 32;
  const int rounds = (n_tiles + kTopWaves - 1) / kTopWaves;
  for (int rd = 0; rd < rounds; rd++) {
    const int tl = rd * kTopWaves + wv;
    if (tl < n_tiles) {
      const int i0 = tl * 32;
      // stage the item tile: coalesced reads of 32 consecutive item vectors
      for (int e = lane; e < 32 * k_rank; e += 64) {
        const int it = e / k_rank, kk = e - it * k_rank;
        tile[it * LDT + kk] = (i0 + it < n_items) ? V[(size_t)(i0 + it) * k_rank + kk] : 0.f;
      }
      wave_sync();
      f32x16_t acc;
#pragma unroll
      for (int e = 0; e < 16; e++) acc[e] = 0.f;
#pragma unroll
      for (int t = 0; t < NK2; t++) {
        const float b = tile[col * LDT + 2 * t + half];  // B[kk = 2t + half][item = col]
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[t], b, acc, 0, 0, 0);
      }
      wave_sync();
      // lane holds item `col` for users row(e) = (e & 3) + 8 (e >> 2) + 4 half
      const int item = i0 + col;
      const bool item_ok = item < n_items && !(n_excl > 0 && sorted_contains(excl, n_excl, item));
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int ul = (e & 3) + 8 * (e >> 2) + 4 * half;
        const int u = u0 + ul;
        const float s = acc[e];
        if (item_ok && u < n_users && s > sThr[ul]) {
          bool skip = false;
          if (nr_ptr) {
            const int p1 = nr_ptr[u], p2 = nr_ptr[u + 1];
            skip = sorted_contains(nr_idx + p1, p2 - p1, item);
          }
          if (!skip) {
            const int pos = atomicAdd(&sCnt[ul], 1);
            if (pos < kTopCap) {
              sVal[ul * kTopCap + pos] = s;
              sIdx[ul * kTopCap + pos] = item;
            }
          }
        }
      }
    }
    __syncthreads();
    // reduce over-full buffers to their top k (one wave per user, rank counting), raise the threshold
    for (int ul = wv; ul < kTopUsers; ul += kTopWaves) {
      const int n = min(sCnt[ul], kTopCap);
      if (n > topk) {  // wave-uniform
        float* bv = sVal + ul * kTopCap;
        int* bi = sIdx + ul * kTopCap;
        float* tv = sTmpV + wv * kTopCap;
        int* ti = sTmpI + wv * kTopCap;
        for (int c = lane; c < n; c += 64) {
          const float v = bv[c];
          const int ix = bi[c];
          int rank = 0;
          for (int c2 = 0; c2 < n; c2++) rank += sel_before(bv[c2], bi[c2], v, ix) ? 1 : 0;
          if (rank < topk) {
            tv[rank] = v;
            ti[rank] = ix;
          }
        }
        wave_sync();
        // More candidates at the k-th score than places for them?  Then WHICH of them survive depends on the order in
        // which the reference's heap met them (it evicts the smallest index among tied minima, but a tied newcomer
        // never enters a full heap: scores [1,1,5], k = 2 -> {2,1}; [1,5,1] -> {1,0}), and the arrivals since the last
        // reduction are replayed through that heap in index order.  Everything that arrived before was reduced the
        // same way, so the buffer's first sNeed entries ARE the reference's heap at that point.
        const float kth = tv[topk - 1];
        int ge = 0;
        for (int c = lane; c < n; c += 64) ge += bv[c] >= kth ? 1 : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ge += __shfl_xor(ge, off);
        if (ge > topk) {   // wave-uniform
          const int old = sNeed[ul];
          wave_sync();
          for (int c = lane; c < n; c += 64) {   // heap as it is, then the new arrivals sorted by item index
            int dst = c;
            if (c >= old) {
              const int ix = bi[c];
              int r = 0;
              for (int c2 = old; c2 < n; c2++) r += bi[c2] < ix ? 1 : 0;
              dst = old + r;
            }
            tv[dst] = bv[c];
            ti[dst] = bi[c];
          }
          wave_sync();
          int h = old;
          for (int c = old; c < n; c++) {
            const float v = tv[c];
            const int ix = ti[c];
            if (h < topk) {
              if (lane == 0) {
                bv[h] = v;
                bi[h] = ix;
              }
              h++;
            } else {
              // smallest (score, index) of the heap: what std::priority_queue<pair, greater> has on top
              float mv = INFINITY;
              int mi = 0x7fffffff, mp = -1;
              for (int e = lane; e < topk; e += 64) {
                const float hv = bv[e];
                const int hi = bi[e];
                if (hv < mv || (hv == mv && hi < mi)) { mv = hv; mi = hi; mp = e; }
              }
#pragma unroll
              for (int off = 32; off > 0; off >>= 1) {
                const float ov = __shfl_xor(mv, off);
                const int oi = __shfl_xor(mi, off), op = __shfl_xor(mp, off);
                if (ov < mv || (ov == mv && oi < mi)) { mv = ov; mi = oi; mp = op; }
              }
              if (mv < v && lane == 0) {
                bv[mp] = v;
                bi[mp] = ix;
              }
            }
            wave_sync();
          }
          float mn = INFINITY;
          for (int e = lane; e < topk; e += 64) mn = fminf(mn, bv[e]);
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) mn = fminf(mn, __shfl_xor(mn, off));
          if (lane == 0) {
            sCnt[ul] = topk;
            sThr[ul] = mn;
            sNeed[ul] = topk;
          }
        } else {
          for (int c = lane; c < topk; c += 64) {
            bv[c] = tv[c];
            bi[c] = ti[c];
          }
          wave_sync();
          if (lane == 0) {
            sCnt[ul] = topk;
            sThr[ul] = kth;
            sNeed[ul] = topk;
          }
        }
      }
    }
    __syncthreads();
  }
  // output: best first; equal scores with the larger index first (heap pop order of the reference)
  for (int ul = wv; ul < kTopUsers; ul += kTopWaves) {
    const int u = u0 + ul;
    if (u >= n_users) continue;
    const int n = min(min(sCnt[ul], kTopCap), topk);
    const float* bv = sVal + ul * kTopCap;
    const int* bi = sIdx + ul * kTopCap;
    for (int c = lane; c < topk; c += 64) {
      if (c < n) {
        const float v = bv[c];
        const int ix = bi[c];
        int rank = 0;
        for (int c2 = 0; c2 < n; c2++) {
          const float v2 = bv[c2];
          rank += (v2 > v || (v2 == v && bi[c2] > ix)) ? 1 : 0;
        }
        res[(size_t)u * topk + rank] = ix + 1;               // 1-based, like R
        scores_out[(size_t)u * topk + rank] = v + glob_mean;
      } else {
        res[(size_t)u * topk + c] = INT32_MIN;                // NA_integer_
        scores_out[(size_t)u * topk + c] = __int_as_float(0x7fc00000);
      }
    }
  }
}

}  // namespace

size_t top_product_lds_bytes(int k_rank) {
  const int KP = padded_rank(k_rank);
  if (KP == 32) return TopSmem<32>::bytes;
  if (KP == 64) return TopSmem<64>::bytes;
  return TopSmem<128>::bytes;
}

hipError_t launch_top_product(const float* U, const float* V, int n_users, int n_items, int k_rank, int topk,
                              const int32_t* nr_ptr, const int32_t* nr_idx, const int32_t* excl, int n_excl,
                              float glob_mean, int32_t* res, float* scores, hipStream_t s) {
  const int KP = padded_rank(k_rank);
  if (!KP || topk < 1 || topk > kTopCap - 128) return hipErrorInvalidValue;
  if (n_users <= 0) return hipSuccess;
  const int grid = (n_users + kTopUsers - 1) / kTopUsers;
  hipError_t err;
#define RSP_TOPK(KPV)                                                                                        \
  if (KP == KPV) {                                                                                           \
    auto kern = top_product_kernel<KPV>;                                                                     \
    if ((err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   (int)TopSmem<KPV>::bytes)) != hipSuccess)                                 \
      return err;                                                                                            \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), TopSmem<KPV>::bytes, s, U, V, n_users, n_items, k_rank, topk, \
                       nr_ptr, nr_idx, excl, n_excl, glob_mean, res, scores);                                \
    return hipGetLastError();                                                                                \
  }
  RSP_TOPK(32)
  RSP_TOPK(64)
  RSP_TOPK(128)
#undef RSP_TOPK
  return hipErrorInvalidValue;
}

}  // namespace rsparse_hip
