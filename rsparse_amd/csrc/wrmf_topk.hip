// top-k of the dense product  scores = U * V^T  per user, fused (gfx950, wave64).
//
// Replaces top_product() (src/matrix_top_product.cpp:20-102), the C++ behind find_top_product()
// (R/utils.R:31-59) and `$predict` (R/MatrixFactorizationRecommender.R:24-78): for every user row j,
// scores_i = u_j . v_i over all items, skip the items of the user's `not_recommend` row (sorted CSR) and the
// globally excluded items, keep the k best in a min-heap (`q.top().first < val` -> on ties the earlier item
// stays), emit them best first -- equal scores come out with the larger index first, because the heap pops
// (score, index) pairs in ascending pair order and the output is filled from the end (:88-95).
//
// (What follows describes the first geometry -- a wave per item tile, the candidate buffers in LDS --, which still serves calls
// for up to 32 / 64 users.  Calls for more than 128 users run top_product_pipe_kernel<.., GBUF>: the four waves share the item
// tile and own 64 users each, two tiles resident, the candidate buffers in a global scratch and settled once per batch of
// arrivals by a radix select; few users over many items are split over the items and merged.  See launch_top_product_geo.)
// One 256-thread workgroup per block of 32 * UB users (UB = 2 when the candidate buffers fit LDS, i.e. k <= 16).  The
// user blocks are the MFMA A operands and stay in registers (UB * k/2 VGPRs); each of the 4 waves walks its own 32-item
// tiles: while the matrix cores work on tile n, the 16-byte coalesced loads of tile n + 1 are in flight into registers
// (one wave per SIMD: nothing else would hide them), then the tile goes to LDS (row stride k + 4: aligned 16-byte
// stores; the 4-way bank conflict of the fragment reads is invisible next to a 64-cycle MFMA), is read back as B
// fragments -- each fragment feeds UB MFMAs -- and multiplied with v_mfma_f32_32x32x2_f32 (exact fp32).  A score survives
// only if it beats the user's current k-th best (threshold in LDS); survivors that pass the exclusion checks (binary
// searches) are appended to the user's LDS buffer; once per round of 4 tiles, buffers holding more than k entries are
// reduced to their top k by rank counting and the threshold is raised.  After the first few tiles almost nothing
// survives.  (Round 3: the staging loop used to be 4-byte loads with a run-time division per element and no load in
// flight during the MFMAs -- 8.6 TFLOP/s at 1M x 1M, rank 128; see profiles/r03.)
#include <algorithm>

#include "wrmf_internal.h"
#include "wrmf_device.h"

namespace rsparse_hip {
namespace {

using namespace dev;

typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr int kTopBlock = 32;    // users per MFMA block
constexpr int kTopMaxK = 256;    // RSPARSE_HIP_MAX_TOPK
constexpr int kTopMaxCap = 576;  // no candidate buffer is longer (top_gcap(256) = 544; the LDS geometries: k + 128 at most)
// candidate buffer per user: the heap (k entries) + everything one round can add (32 items per tile of the round)
__host__ __device__ constexpr int top_cap(int topk, int tiles_per_round) { return ((topk + 32 * tiles_per_round + 3) / 4) * 4; }

// GBUF (round 6): the candidate buffers of 256 users fit the LDS next to two tiles only up to k = 27 at rank 128, and only without
// a spare entry: a user was settled at EVERY arrival.  The tile-sharing kernel now keeps them in GLOBAL memory (a scratch slot
// per workgroup: appends are fire-and-forget stores, only the counts and thresholds stay in LDS) with room for a BATCH of
// arrivals beyond one tile's worth -- a user is settled once per max(64, k) arrivals, in the settling wave's LDS work area,
// by a radix select (topk_reduce_user).  One geometry, 256 users per workgroup, for every k.
#ifdef RSP_TOPK_GBATCH   // dev builds: a fixed batch room
__host__ __device__ constexpr int top_gcap(int topk) { return ((topk + 32 + RSP_TOPK_GBATCH + 3) / 4) * 4; }
#else
__host__ __device__ constexpr int top_gcap(int topk) { return ((topk + 32 + (topk > 64 ? topk : 64) + 3) / 4) * 4; }
#endif
constexpr int kTopGbufChunk = 131072;   // users per launch of a GBUF call: 512 slots of scratch, two full rounds of workgroups

// the kernel in which every wave walks its own item tiles against the workgroup's 32 UB users (W waves = W tiles per round)
template <int KP, int UB, int W>
struct TopSmem {
  static constexpr int LDT = KP + 4;  // rows 16-byte aligned (ds_write_b128); fragment reads: 4-way conflict, hidden
  static constexpr int USERS = kTopBlock * UB;
  static constexpr size_t tile_floats = (size_t)W * 32 * LDT;
  static size_t bytes(int topk) {
    const size_t cap = (size_t)top_cap(topk, W);
    return (tile_floats + 2 * (size_t)USERS * cap + 8 * USERS) * 4 + 64 + (size_t)W * cap * 8;   // + compaction scratch
  }
};
// the kernel in which the four waves share ONE item tile and own 32 UB users each (round 4)
template <int KP, int UB>
struct TopSharedSmem {
  static constexpr int LDT = KP + 4;
  static constexpr int USERS = 4 * kTopBlock * UB;
  static constexpr size_t tile_floats = (size_t)32 * LDT;
  static size_t bytes(int topk) {
    const size_t cap = (size_t)top_cap(topk, 1);
    return (tile_floats + 2 * (size_t)USERS * cap + 3 * USERS) * 4 + 64 + (size_t)4 * cap * 8;
  }
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

// Reduce one user's over-full candidate buffer (n > topk entries: bv / bi) to its top k by rank counting (tv / ti: scratch
// of the calling wave), and return the new threshold (the k-th best score).  `old` = how many leading entries are the heap
// as the previous reduction left it.  One wave; every lane gets the result.
__device__ __forceinline__ float topk_reduce_user(float* bv, int* bi, float* tv, int* ti, const int n, const int topk,
                                                  const int old, const int lane) {
  // A full heap and a handful of arrivals (the steady state: after the first tiles a user sees a survivor every few hundred
  // tiles): the arrivals go through the heap one by one -- the reference's own loop, src/matrix_top_product.cpp:61-86, and
  // the code below for the tie case -- at O(k / 64) each, instead of the O(n^2 / 64) rank counting that pays for big batches.
  const bool few = old == topk && n - old <= 8;
  float kth = 0.f;
  int ge = topk + 1;
  // (round 6) The k-th best score by a radix select over the entries held in registers -- 32 bit positions x a ballot per register
  // -- instead of ranking every entry against every other (O(n^2 / 64): 15 us at n = 220, and the batched settles of the global
  // buffers made it the whole cost of top-100).  The buffer's order carries no meaning (the heap is a set: the replay below scans
  // it for its minimum, the emission ranks it), so the survivors -- the entries at or above the k-th score, exactly k of them
  // unless the bound is tied -- are compacted to the front in place.
  constexpr int NE = (kTopMaxCap + 63) / 64;
  unsigned key[NE];
  int eix[NE];
  if (!few) {
#pragma unroll
    for (int j = 0; j < NE; j++) {
      const int c = lane + 64 * j;
      key[j] = 0u;   // (below every score's key: a finite float maps to a key >= 0x00800000)
      eix[j] = 0;
      if (c < n) {
        const unsigned u = __float_as_uint(bv[c] + 0.f);   // (-0 -> +0: equal scores, equal keys)
        key[j] = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        eix[j] = bi[c];
      }
    }
    unsigned kk = 0u;   // the k-th largest key, bit by bit
    for (int bit = 31; bit >= 0; bit--) {
      const unsigned trial = kk | (1u << bit);
      int cnt = 0;
#pragma unroll
      for (int j = 0; j < NE; j++)
        if (64 * j < n) cnt += __popcll(__ballot(key[j] >= trial));
      if (cnt >= topk) kk = trial;
    }
    ge = 0;
#pragma unroll
    for (int j = 0; j < NE; j++)
      if (64 * j < n) ge += __popcll(__ballot(key[j] >= kk));
    kth = __uint_as_float((kk & 0x80000000u) ? (kk & 0x7fffffffu) : ~kk);
    // More candidates at the k-th score than places for them?  Then WHICH of them survive depends on the order in
    // which the reference's heap met them (it evicts the smallest index among tied minima, but a tied newcomer
    // never enters a full heap: scores [1,1,5], k = 2 -> {2,1}; [1,5,1] -> {1,0}), and the arrivals since the last
    // reduction are replayed through that heap in index order.  Everything that arrived before was reduced the
    // same way, so the buffer's first `old` entries ARE the reference's heap at that point.
    if (ge == topk) {
      wave_sync();   // (every lane holds its entries; the buffer is rewritten in place)
      int base = 0;
#pragma unroll
      for (int j = 0; j < NE; j++) {
        if (64 * j < n) {
          const bool keep = key[j] >= kk;
          const unsigned long long m = __ballot(keep);
          if (keep) {
            const int dst = base + __popcll(m & ((1ull << lane) - 1ull));
            const unsigned kj = key[j];
            bv[dst] = __uint_as_float((kj & 0x80000000u) ? (kj & 0x7fffffffu) : ~kj);
            bi[dst] = eix[j];
          }
          base += __popcll(m);
        }
      }
      wave_sync();
      return kth;
    }
  }
  if (ge > topk) {   // wave-uniform
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
    return mn;
  }
  return kth;   // (not reached: ge >= topk whenever n > topk)
}

// The tile-sharing kernels append a score that beats its user's threshold WITHOUT looking at the exclusion lists (the
// unrolled epilogue stays a compare and an append per accumulator entry: with two binary searches inlined 32 times the loop
// outgrew the instruction cache); the lists are consulted here, one lane per arrival, before a buffer is reduced or emitted.
// Drops the entries [from, n) of a user's buffer that `nr_row` (the user's not_recommend row) or `excl` rules out, closes
// the gaps, returns the new count.  One wave.
__device__ __forceinline__ int topk_verify_arrivals(float* bv, int* bi, const int from, const int n,
                                                    const int32_t* __restrict__ nr_row, const int nr_len,
                                                    const int32_t* __restrict__ excl, const int n_excl, const int lane) {
  if (nr_len <= 0 && n_excl <= 0) return n;
  int kept = from;
  for (int c0 = from; c0 < n; c0 += 64) {
    const int c = c0 + lane;
    float v = 0.f;
    int ix = 0;
    bool ok = false;
    if (c < n) {
      v = bv[c];
      ix = bi[c];
      ok = !(nr_len > 0 && sorted_contains(nr_row, nr_len, ix)) && !(n_excl > 0 && sorted_contains(excl, n_excl, ix));
    }
    const unsigned long long m = __ballot(ok);
    wave_sync();   // every lane has read its entry before any is overwritten
    if (ok) {
      const int dst = kept + __popcll(m & ((1ull << lane) - 1ull));
      bv[dst] = v;
      bi[dst] = ix;
    }
    kept += __popcll(m);
    wave_sync();
  }
  return kept;
}

// arrivals verified, then the buffer reduced if it is over-full: what the tile-sharing kernels do for a user whose count
// passed k.  Updates the user's count / threshold / verified length.  One wave.
__device__ __forceinline__ void topk_settle_user(float* bv, int* bi, float* tv, int* ti, int* cnt, float* thr, int* need,
                                                 const int topk, const int32_t* __restrict__ nr_ptr,
                                                 const int32_t* __restrict__ nr_idx, const int u,
                                                 const int32_t* __restrict__ excl, const int n_excl, const int lane) {
  const int old = *need;
  int p1 = 0, len = 0;
  if (nr_ptr) {
    p1 = nr_ptr[u];
    len = nr_ptr[u + 1] - p1;
  }
  const int n = topk_verify_arrivals(bv, bi, old, *cnt, nr_idx + p1, len, excl, n_excl, lane);
  float t = *thr;
  int nn = n;
  if (n > topk) {   // wave-uniform
    t = topk_reduce_user(bv, bi, tv, ti, n, topk, old, lane);
    nn = topk;
  }
  wave_sync();
  if (lane == 0) {
    *cnt = nn;
    *thr = t;
    *need = nn;
  }
  wave_sync();
}

// final output of one user: best first; equal scores with the larger index first (heap pop order of the reference)
__device__ __forceinline__ void topk_emit_user(const float* bv, const int* bi, const int n, const int topk, const int lane,
                                               const float glob_mean, int32_t* res_u, float* scores_u) {
  for (int c = lane; c < topk; c += 64) {
    if (c < n) {
      const float v = bv[c];
      const int ix = bi[c];
      int rank = 0;
      for (int c2 = 0; c2 < n; c2++) {
        const float v2 = bv[c2];
        rank += (v2 > v || (v2 == v && bi[c2] > ix)) ? 1 : 0;
      }
      res_u[rank] = ix + 1;               // 1-based, like R
      scores_u[rank] = v + glob_mean;
    } else {
      res_u[c] = INT32_MIN;                // NA_integer_
      scores_u[c] = __int_as_float(0x7fc00000);
    }
  }
}

// A split launch (few users, many items: blockIdx.y = item slice): the workgroup sees its slice as the matrix -- item ids are
// item_base + local -- and writes to the slice's block of the scratch; the fall-back launch of the unsplit kernel (user_flags
// given) runs only the workgroups that hold a flagged user.  See launch_top_product.
#define RSP_TOPK_SLICE(USERS_, THREADS_)                                                                 \
  int item_base = 0;                                                                                     \
  if (slice_items > 0) {                                                                                 \
    item_base = (int)blockIdx.y * slice_items;                                                           \
    V += (size_t)item_base * k_rank;                                                                     \
    n_items = max(0, min(n_items - item_base, slice_items));                                             \
    res += (size_t)blockIdx.y * n_users * topk;                                                          \
    scores_out += (size_t)blockIdx.y * n_users * topk;                                                   \
  }                                                                                                      \
  if (user_flags) {                                                                                      \
    int f_ = 0;                                                                                          \
    for (int e_ = threadIdx.x; e_ < (USERS_); e_ += (THREADS_))                                          \
      if (u0 + e_ < n_users) f_ |= user_flags[u0 + e_];                                                  \
    if (!__syncthreads_or(f_)) return;                                                                   \
  }

template <int KP, int UB, bool VEC, int W>
__global__ __launch_bounds__(W * 64) void top_product_kernel(const float* __restrict__ U, const float* __restrict__ V,
                                                          int n_users, int n_items, int k_rank, int topk,
                                                          const int32_t* __restrict__ nr_ptr,
                                                          const int32_t* __restrict__ nr_idx,
                                                          const int32_t* __restrict__ excl, int n_excl,
                                                          float glob_mean, int32_t* __restrict__ res,
                                                          float* __restrict__ scores_out, int slice_items,
                                                          const int* __restrict__ user_flags) {
  using SM = TopSmem<KP, UB, W>;
  constexpr int LDT = SM::LDT, NK2 = KP / 2, kTopUsers = SM::USERS, kTopWaves = W;
  const int kTopCap = top_cap(topk, W);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sTile = reinterpret_cast<float*>(smem);
  float* sVal = sTile + SM::tile_floats;                            // [USERS][CAP]
  int* sIdx = reinterpret_cast<int*>(sVal + (size_t)kTopUsers * kTopCap);   // [USERS][CAP]
  int* sCnt = sIdx + (size_t)kTopUsers * kTopCap;                   // [USERS]
  float* sThr = reinterpret_cast<float*>(sCnt + kTopUsers);         // [USERS]
  int* sNeed = reinterpret_cast<int*>(sThr + kTopUsers);            // [USERS] spare / flags
  float* sTmpV = reinterpret_cast<float*>(sNeed + 2 * kTopUsers + 4);  // [W][CAP]
  int* sTmpI = reinterpret_cast<int*>(sTmpV + kTopWaves * kTopCap);    // [W][CAP]

  const int tid = threadIdx.x, lane = tid & 63, wv = rfl(tid >> 6);
  const int u0 = blockIdx.x * kTopUsers;
  RSP_TOPK_SLICE(kTopUsers, W * 64)
  const int col = lane & 31, half = lane >> 5;
  if (tid < kTopUsers) {
    sCnt[tid] = 0;
    sThr[tid] = -INFINITY;
    sNeed[tid] = 0;   // entries [0, sNeed) of the user's buffer are the heap as the last reduction left it
  }
  // A operands: lane holds U[u0 + 32 ub + (lane & 31)][2t + half], t = 0..KP/2-1 (zero beyond the matrix)
  float afrag[UB][NK2];
#pragma unroll
  for (int ub = 0; ub < UB; ub++) {
    const int u = u0 + 32 * ub + col;
#pragma unroll
    for (int t = 0; t < NK2; t++) {
      const int kk = 2 * t + half;
      afrag[ub][t] = (u < n_users && kk < k_rank) ? U[(size_t)u * k_rank + kk] : 0.f;
    }
  }
  float* tile = sTile + wv * 32 * LDT;
  for (int e = lane; e < 32 * LDT; e += 64) tile[e] = 0.f;
  __syncthreads();

  const int n_tiles = (n_items + 31) / 32;
  const int rounds = (n_tiles + kTopWaves - 1) / kTopWaves;
  // item tile tl -> registers: VEC = rank a multiple of 4 and V 16-byte aligned: NLD 16-byte loads per lane, all in
  // flight together (piece e4 = j * 64 + lane: item e4 / (KP / 4), floats 4 (e4 % (KP / 4)) ..+3)
  constexpr int NLD = KP / 8;
  float4 pf[VEC ? NLD : 1];
  auto load_tile = [&](const int tl) {
    if constexpr (VEC) {
      const int i0 = tl * 32;
#pragma unroll
      for (int j = 0; j < NLD; j++) {
        const int e4 = j * 64 + lane, it = e4 / (KP / 4), c4 = e4 % (KP / 4);
        pf[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tl < n_tiles && i0 + it < n_items && 4 * c4 < k_rank)
          pf[j] = *reinterpret_cast<const float4*>(V + (size_t)(i0 + it) * k_rank + 4 * c4);
      }
    }
  };
  auto store_tile = [&](const int tl) {
    if constexpr (VEC) {
#pragma unroll
      for (int j = 0; j < NLD; j++) {
        const int e4 = j * 64 + lane, it = e4 / (KP / 4), c4 = e4 % (KP / 4);
        *reinterpret_cast<float4*>(tile + it * LDT + 4 * c4) = pf[j];
      }
    } else {   // any rank / alignment: scalar staging (slow path)
      const int i0 = tl * 32;
      for (int it = 0; it < 32; it++)
        for (int kk = lane; kk < k_rank; kk += 64)
          tile[it * LDT + kk] = (tl < n_tiles && i0 + it < n_items) ? V[(size_t)(i0 + it) * k_rank + kk] : 0.f;
    }
  };
  load_tile(wv);
  store_tile(wv);
  wave_sync();
  for (int rd = 0; rd < rounds; rd++) {
    const int tl = rd * kTopWaves + wv;
    load_tile(tl + kTopWaves);   // the next tile's loads fly during this tile's MFMAs
    if (tl < n_tiles) {
      const int i0 = tl * 32;
      f32x16_t acc[UB];
#pragma unroll
      for (int ub = 0; ub < UB; ub++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[ub][e] = 0.f;
#pragma unroll
      for (int t = 0; t < NK2; t++) {
        const float b = tile[col * LDT + 2 * t + half];  // B[kk = 2t + half][item = col]
#pragma unroll
        for (int ub = 0; ub < UB; ub++) acc[ub] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[ub][t], b, acc[ub], 0, 0, 0);
      }
      // lane holds item `col` for users row(e) = (e & 3) + 8 (e >> 2) + 4 half of each block
      const int item = i0 + col, gitem = item + item_base;
      const bool item_ok = item < n_items && !(n_excl > 0 && sorted_contains(excl, n_excl, gitem));
#pragma unroll
      for (int ub = 0; ub < UB; ub++)
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int ul = 32 * ub + (e & 3) + 8 * (e >> 2) + 4 * half;
          const int u = u0 + ul;
          const float s = acc[ub][e];
          if (item_ok && u < n_users && s > sThr[ul]) {
            bool skip = false;
            if (nr_ptr) {
              const int p1 = nr_ptr[u], p2 = nr_ptr[u + 1];
              skip = sorted_contains(nr_idx + p1, p2 - p1, gitem);
            }
            if (!skip) {
              const int pos = atomicAdd(&sCnt[ul], 1);
              if (pos < kTopCap) {
                sVal[ul * kTopCap + pos] = s;
                sIdx[ul * kTopCap + pos] = gitem;
              }
            }
          }
        }
    }
    wave_sync();                 // this wave has read its tile for the last time
    store_tile(tl + kTopWaves);
    __syncthreads();
    // reduce over-full buffers to their top k (one wave per user, rank counting), raise the threshold
    for (int ul = wv; ul < kTopUsers; ul += kTopWaves) {
      const int n = min(sCnt[ul], kTopCap);
      if (n > topk) {  // wave-uniform
        const float thr = topk_reduce_user(sVal + ul * kTopCap, sIdx + ul * kTopCap, sTmpV + wv * kTopCap, sTmpI + wv * kTopCap, n,
                                           topk, sNeed[ul], lane);
        if (lane == 0) {
          sCnt[ul] = topk;
          sThr[ul] = thr;
          sNeed[ul] = topk;
        }
      }
    }
    __syncthreads();
  }
  for (int ul = wv; ul < kTopUsers; ul += kTopWaves) {
    const int u = u0 + ul;
    if (u >= n_users) continue;
    topk_emit_user(sVal + ul * kTopCap, sIdx + ul * kTopCap, min(min(sCnt[ul], kTopCap), topk), topk, lane, glob_mean,
                   res + (size_t)u * topk, scores_out + (size_t)u * topk);
  }
}


// Round 4: the four waves SHARE one item tile and own 32 UB users each (128 / 256 users per workgroup).  The kernel above
// reads every item vector once per 32 UB users -- at 1M x 1M, rank 128 that is 8 TB through the L2 for 64 users per
// workgroup, and the matrix cores wait for it (52.7 TFLOP/s; one user block, as top-100 needs there, 9) --, this one once per
// 128 UB: the tile is staged cooperatively (the next tile's 16-byte loads fly during this tile's MFMAs, as above), every
// wave multiplies it with its own user blocks, and because a wave owns its users' candidate buffers outright the buffers
// need room for ONE tile's survivors only (cap = k + 32) and are reduced by their wave without a workgroup barrier.
template <int KP, int UB, bool VEC>
__global__ __launch_bounds__(256) void top_product_shared_kernel(const float* __restrict__ U, const float* __restrict__ V,
                                                                 int n_users, int n_items, int k_rank, int topk,
                                                                 const int32_t* __restrict__ nr_ptr,
                                                                 const int32_t* __restrict__ nr_idx,
                                                                 const int32_t* __restrict__ excl, int n_excl,
                                                                 float glob_mean, int32_t* __restrict__ res,
                                                                 float* __restrict__ scores_out, int slice_items,
                                                                 const int* __restrict__ user_flags) {
  using SM = TopSharedSmem<KP, UB>;
  constexpr int LDT = SM::LDT, NK2 = KP / 2, USERS = SM::USERS, UPW = 32 * UB;   // users per wave
  const int cap = top_cap(topk, 1);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tile = reinterpret_cast<float*>(smem);                     // [32][LDT]
  float* sVal = tile + SM::tile_floats;                             // [USERS][cap]
  int* sIdx = reinterpret_cast<int*>(sVal + (size_t)USERS * cap);   // [USERS][cap]
  int* sCnt = sIdx + (size_t)USERS * cap;                           // [USERS]
  float* sThr = reinterpret_cast<float*>(sCnt + USERS);             // [USERS]
  int* sNeed = reinterpret_cast<int*>(sThr + USERS);                // [USERS]
  float* sTmpV = reinterpret_cast<float*>(sNeed + USERS);           // [4][cap]
  int* sTmpI = reinterpret_cast<int*>(sTmpV + 4 * cap);             // [4][cap]

  const int tid = threadIdx.x, lane = tid & 63, wv = rfl(tid >> 6);
  const int u0 = blockIdx.x * USERS, uw = wv * UPW;                 // this wave's users: [u0 + uw, u0 + uw + UPW)
  RSP_TOPK_SLICE(USERS, 256)
  const int col = lane & 31, half = lane >> 5;
  for (int e = tid; e < USERS; e += 256) {
    sCnt[e] = 0;
    sThr[e] = -INFINITY;
    sNeed[e] = 0;
  }
  float afrag[UB][NK2];
#pragma unroll
  for (int ub = 0; ub < UB; ub++) {
    const int u = u0 + uw + 32 * ub + col;
#pragma unroll
    for (int t = 0; t < NK2; t++) {
      const int kk = 2 * t + half;
      afrag[ub][t] = (u < n_users && kk < k_rank) ? U[(size_t)u * k_rank + kk] : 0.f;
    }
  }
  for (int e = tid; e < 32 * LDT; e += 256) tile[e] = 0.f;
  __syncthreads();

  const int n_tiles = (n_items + 31) / 32;
  constexpr int NLD = KP / 32;   // 16-byte pieces per thread and tile
  float4 pf[VEC ? NLD : 1];
  auto load_tile = [&](const int tl) {
    if constexpr (VEC) {
      const int i0 = tl * 32;
#pragma unroll
      for (int j = 0; j < NLD; j++) {
        const int e4 = j * 256 + tid, it = e4 / (KP / 4), c4 = e4 % (KP / 4);
        pf[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tl < n_tiles && i0 + it < n_items && 4 * c4 < k_rank)
          pf[j] = *reinterpret_cast<const float4*>(V + (size_t)(i0 + it) * k_rank + 4 * c4);
      }
    }
  };
  auto store_tile = [&](const int tl) {
    if constexpr (VEC) {
#pragma unroll
      for (int j = 0; j < NLD; j++) {
        const int e4 = j * 256 + tid, it = e4 / (KP / 4), c4 = e4 % (KP / 4);
        *reinterpret_cast<float4*>(tile + it * LDT + 4 * c4) = pf[j];
      }
    } else {   // any rank / alignment: scalar staging (slow path)
      const int i0 = tl * 32;
      for (int it = wv; it < 32; it += 4)
        for (int kk = lane; kk < k_rank; kk += 64)
          tile[it * LDT + kk] = (tl < n_tiles && i0 + it < n_items) ? V[(size_t)(i0 + it) * k_rank + kk] : 0.f;
    }
  };
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int tl = 0; tl < n_tiles; tl++) {
    load_tile(tl + 1);   // the next tile's loads fly during this tile's MFMAs
    const int i0 = tl * 32;
    f32x16_t acc[UB];
#pragma unroll
    for (int ub = 0; ub < UB; ub++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[ub][e] = 0.f;
#pragma unroll
    for (int t = 0; t < NK2; t++) {
      const float b = tile[col * LDT + 2 * t + half];  // B[kk = 2t + half][item = col]
#pragma unroll
      for (int ub = 0; ub < UB; ub++) acc[ub] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[ub][t], b, acc[ub], 0, 0, 0);
    }
    // lane holds item `col` for users row(e) = (e & 3) + 8 (e >> 2) + 4 half of each block
    const int item = i0 + col;
    const bool item_ok = item < n_items;
    // the thresholds of this lane's users, four 16-byte reads per block in flight together (one read and one wait per
    // accumulator entry left the matrix cores idle for a third of the tile)
    float4 thr4[UB][4];
#pragma unroll
    for (int ub = 0; ub < UB; ub++)
#pragma unroll
      for (int q = 0; q < 4; q++) thr4[ub][q] = *reinterpret_cast<const float4*>(sThr + uw + 32 * ub + 8 * q + 4 * half);
#pragma unroll
    for (int ub = 0; ub < UB; ub++)
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int ul = uw + 32 * ub + (e & 3) + 8 * (e >> 2) + 4 * half;
        const float sc = acc[ub][e];
        const float* tq = reinterpret_cast<const float*>(&thr4[ub][e >> 2]);
        if (item_ok && u0 + ul < n_users && sc > tq[e & 3]) {   // appended unverified: topk_settle_user consults the lists
          const int pos = atomicAdd(&sCnt[ul], 1);               // (lanes of this wave only: at most 32 per user and tile)
          sVal[ul * cap + pos] = sc;
          sIdx[ul * cap + pos] = item + item_base;
        }
      }
    wave_sync();
    // this wave's over-full buffers: arrivals verified, top k kept, threshold raised (after the first tiles almost nothing survives)
    for (int b0 = 0; b0 < UPW; b0 += 64) {
      const int ulq = uw + b0 + lane;
      unsigned long long over = __ballot(b0 + lane < UPW && sCnt[ulq] > topk);
      while (over) {
        const int ul = uw + b0 + __builtin_ctzll(over);
        over &= over - 1;
        topk_settle_user(sVal + ul * cap, sIdx + ul * cap, sTmpV + wv * cap, sTmpI + wv * cap, sCnt + ul, sThr + ul, sNeed + ul,
                         topk, nr_ptr, nr_idx, u0 + ul, excl, n_excl, lane);
      }
    }
    __syncthreads();             // every wave has read the tile for the last time
    store_tile(tl + 1);
    __syncthreads();
  }
  for (int ul = uw; ul < uw + UPW; ul++) {
    const int u = u0 + ul;
    if (u >= n_users) break;
    if (sCnt[ul] > sNeed[ul])   // arrivals that no reduction has looked at yet (the buffer never filled up again)
      topk_settle_user(sVal + ul * cap, sIdx + ul * cap, sTmpV + wv * cap, sTmpI + wv * cap, sCnt + ul, sThr + ul, sNeed + ul, topk,
                       nr_ptr, nr_idx, u, excl, n_excl, lane);
    topk_emit_user(sVal + ul * cap, sIdx + ul * cap, min(sCnt[ul], topk), topk, lane, glob_mean, res + (size_t)u * topk,
                   scores_out + (size_t)u * topk);
  }
}

// The same with the epilogue hidden behind the NEXT tile's matrix instructions (round 4).  One wave per SIMD (the user blocks
// take 128 registers, the accumulators 32): nothing else runs while a wave compares its 32 scores per lane with the users'
// thresholds, so in the kernel above the matrix cores idle for that part of every tile.  Here two tiles are resident in LDS
// and two accumulator sets alternate: the MFMAs of tile t + 1 are issued in chunks of KP / 32 k-steps between the
// accumulator entries of tile t (a matrix instruction executes for 64 cycles after it has been issued; the compares, the rare
// appends and the buffer reductions of tile t run meanwhile), and one barrier per tile is enough -- tile t + 2 lands in the
// buffer whose last reader finished before the previous barrier.
template <int KP, int UB>
struct TopPipeSmem {
  static constexpr int LDT = KP + 4;
  static constexpr int USERS = 4 * kTopBlock * UB;
  static constexpr size_t tile_floats = (size_t)2 * 32 * LDT;
  static size_t bytes(int topk) {
    const size_t cap = (size_t)top_cap(topk, 1);
    return (tile_floats + 2 * (size_t)USERS * cap + 3 * USERS) * 4 + 64 + (size_t)4 * cap * 8;
  }
  static size_t gbytes(int topk) {   // GBUF: tiles, counts / thresholds / verified lengths, per wave a work area and a scratch
    return (tile_floats + 3 * USERS) * 4 + 64 + (size_t)4 * top_gcap(topk) * 16;
  }
};

// GBUF: a user's buffer (global: gv / gi, n entries) through the wave's LDS work area (wv_ / wi_): arrivals verified, the buffer
// reduced when it is over-full, what is left written back.  One wave.
__device__ __forceinline__ void topk_settle_user_g(float* gv, int* gi, float* wv_, int* wi_, float* tv, int* ti, int* cnt,
                                                   float* thr, int* need, const int topk, const int32_t* __restrict__ nr_ptr,
                                                   const int32_t* __restrict__ nr_idx, const int u,
                                                   const int32_t* __restrict__ excl, const int n_excl, const int lane) {
  const int n0 = *cnt;
  for (int c = lane; c < n0; c += 64) {
    wv_[c] = gv[c];
    wi_[c] = gi[c];
  }
  wave_sync();
  topk_settle_user(wv_, wi_, tv, ti, cnt, thr, need, topk, nr_ptr, nr_idx, u, excl, n_excl, lane);
  const int nn = *cnt;
  for (int c = lane; c < nn; c += 64) {
    gv[c] = wv_[c];
    gi[c] = wi_[c];
  }
}

template <int KP, int UB, bool VEC, bool GBUF = false>
__global__ __launch_bounds__(256) void top_product_pipe_kernel(const float* __restrict__ U, const float* __restrict__ V,
                                                               int n_users, int n_items, int k_rank, int topk,
                                                               const int32_t* __restrict__ nr_ptr,
                                                               const int32_t* __restrict__ nr_idx,
                                                               const int32_t* __restrict__ excl, int n_excl,
                                                               float glob_mean, int32_t* __restrict__ res,
                                                               float* __restrict__ scores_out, int slice_items,
                                                               const int* __restrict__ user_flags, float* gbuf) {
  using SM = TopPipeSmem<KP, UB>;
  constexpr int LDT = SM::LDT, NK2 = KP / 2, USERS = SM::USERS, UPW = 32 * UB;
  constexpr int CH = NK2 / 16;   // k-steps (x UB matrix instructions) issued per accumulator entry of the previous tile
  const int cap = GBUF ? top_gcap(topk) : top_cap(topk, 1);
  const int settle_at = GBUF ? cap - 32 : topk;   // a user is settled when its count passes this (room for one tile's arrivals)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tiles = reinterpret_cast<float*>(smem);                    // [2][32][LDT]
  // GBUF: the buffers are this workgroup's slot of the global scratch ([USERS][cap] scores, then as many indices)
  float* sVal = GBUF ? gbuf + (size_t)blockIdx.x * 2 * USERS * cap : tiles + SM::tile_floats;
  int* sIdx = reinterpret_cast<int*>(sVal + (size_t)USERS * cap);
  int* sCnt = GBUF ? reinterpret_cast<int*>(tiles + SM::tile_floats) : sIdx + (size_t)USERS * cap;
  float* sThr = reinterpret_cast<float*>(sCnt + USERS);
  int* sNeed = reinterpret_cast<int*>(sThr + USERS);
  float* sTmpV = reinterpret_cast<float*>(sNeed + USERS);
  int* sTmpI = reinterpret_cast<int*>(sTmpV + 4 * cap);
  float* sWrkV = reinterpret_cast<float*>(sTmpI + 4 * cap);          // GBUF: [4][cap] the settling wave's copy of a buffer
  int* sWrkI = reinterpret_cast<int*>(sWrkV + 4 * cap);

  const int tid = threadIdx.x, lane = tid & 63, wv = rfl(tid >> 6);
  const int uw = wv * UPW;
  int item_base = 0;
  if (slice_items > 0) {   // a split launch: this workgroup's slice of the items (see RSP_TOPK_SLICE)
    item_base = (int)blockIdx.y * slice_items;
    V += (size_t)item_base * k_rank;
    n_items = max(0, min(n_items - item_base, slice_items));
    res += (size_t)blockIdx.y * n_users * topk;
    scores_out += (size_t)blockIdx.y * n_users * topk;
  }
  // A workgroup takes the user blocks blockIdx.x, + gridDim.x, ...: one block each in the ordinary launch; the fall-back launch for
  // flagged users (user_flags) comes with a grid of at most 512 and every workgroup walks its blocks' flags -- a workgroup that only
  // finds out that it has nothing to do still costs 6..12 us of a CU that it owns alone (150 KB of LDS, 512 registers per lane),
  // and a grid of one such workgroup per block was 7 % (top-10) / 11 % (top-100) of a call that flags nobody (round 6, rocprof)
  const int n_blocks = (n_users + USERS - 1) / USERS;
  for (int blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
  const int u0 = blk * USERS;
  if (user_flags) {
    int f_ = 0;
    for (int e_ = tid; e_ < USERS; e_ += 256)
      if (u0 + e_ < n_users) f_ |= user_flags[u0 + e_];
    if (!__syncthreads_or(f_)) continue;
  }
  const int col = lane & 31, half = lane >> 5;
  for (int e = tid; e < USERS; e += 256) {
    sCnt[e] = 0;
    sThr[e] = -INFINITY;
    sNeed[e] = 0;
  }
  float afrag[UB][NK2];
#pragma unroll
  for (int ub = 0; ub < UB; ub++) {
    const int u = u0 + uw + 32 * ub + col;
#pragma unroll
    for (int t = 0; t < NK2; t++) {
      const int kk = 2 * t + half;
      afrag[ub][t] = (u < n_users && kk < k_rank) ? U[(size_t)u * k_rank + kk] : 0.f;
    }
  }
  for (int e = tid; e < 2 * 32 * LDT; e += 256) tiles[e] = 0.f;
  __syncthreads();

  const int n_tiles = (n_items + 31) / 32;
  constexpr int NLD = KP / 32;
  float4 pf[VEC ? NLD : 1];
  auto load_tile = [&](const int tl) {
    if constexpr (VEC) {
      const int i0 = tl * 32;
#pragma unroll
      for (int j = 0; j < NLD; j++) {
        const int e4 = j * 256 + tid, it = e4 / (KP / 4), c4 = e4 % (KP / 4);
        pf[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tl < n_tiles && i0 + it < n_items && 4 * c4 < k_rank)
          pf[j] = *reinterpret_cast<const float4*>(V + (size_t)(i0 + it) * k_rank + 4 * c4);
      }
    }
  };
  auto store_tile = [&](const int tl) {
    float* tile = tiles + (tl & 1) * 32 * LDT;
    if constexpr (VEC) {
#pragma unroll
      for (int j = 0; j < NLD; j++) {
        const int e4 = j * 256 + tid, it = e4 / (KP / 4), c4 = e4 % (KP / 4);
        *reinterpret_cast<float4*>(tile + it * LDT + 4 * c4) = pf[j];
      }
    } else {
      const int i0 = tl * 32;
      for (int it = wv; it < 32; it += 4)
        for (int kk = lane; kk < k_rank; kk += 64)
          tile[it * LDT + kk] = (tl < n_tiles && i0 + it < n_items) ? V[(size_t)(i0 + it) * k_rank + kk] : 0.f;
    }
  };
  // tile 0 -> LDS, its products (nothing to hide them behind), tile 1 -> LDS
  load_tile(0);
  store_tile(0);
  load_tile(1);
  __syncthreads();
  f32x16_t acc[2][UB];   // [tile parity]
#pragma unroll
  for (int ub = 0; ub < UB; ub++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[0][ub][e] = 0.f;
#pragma unroll
  for (int t = 0; t < NK2; t++) {
    const float b = tiles[col * LDT + 2 * t + half];
#pragma unroll
    for (int ub = 0; ub < UB; ub++) acc[0][ub] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[ub][t], b, acc[0][ub], 0, 0, 0);
  }
  store_tile(1);
  __syncthreads();

  auto body = [&](const int tl, f32x16_t (&cur)[UB], f32x16_t (&nxt)[UB]) {
    // cur: scores of tile tl; LDS buffer (tl + 1) & 1 holds tile tl + 1
    load_tile(tl + 2);
    const float* tn = tiles + ((tl + 1) & 1) * 32 * LDT + col * LDT + half;
    const int item = tl * 32 + col;
    const bool item_ok = item < n_items;
    float4 thr4[UB][4];
#pragma unroll
    for (int ub = 0; ub < UB; ub++)
#pragma unroll
      for (int q = 0; q < 4; q++) thr4[ub][q] = *reinterpret_cast<const float4*>(sThr + uw + 32 * ub + 8 * q + 4 * half);
#pragma unroll
    for (int ub = 0; ub < UB; ub++)
#pragma unroll
      for (int e = 0; e < 16; e++) nxt[ub][e] = 0.f;
    float bq[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) bq[c] = tn[2 * c];
#pragma unroll
    for (int el = 0; el < 16; el++) {
      // the next tile's k-steps [el CH, el CH + CH) for every user block, then the operands of the chunk after it
#pragma unroll
      for (int c = 0; c < CH; c++)
#pragma unroll
        for (int ub = 0; ub < UB; ub++)
          nxt[ub] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[ub][el * CH + c], bq[c], nxt[ub], 0, 0, 0);
      if (el + 1 < 16) {
#pragma unroll
        for (int c = 0; c < CH; c++) bq[c] = tn[2 * ((el + 1) * CH + c)];
      }
      __builtin_amdgcn_sched_barrier(0);
      // ... and entry el of every user block of the current tile (appended unverified: topk_settle_user consults the lists)
#pragma unroll
      for (int ub = 0; ub < UB; ub++) {
        const int ul = uw + 32 * ub + (el & 3) + 8 * (el >> 2) + 4 * half;
        const float sc = cur[ub][el];
        const float* tq = reinterpret_cast<const float*>(&thr4[ub][el >> 2]);
        if (item_ok && u0 + ul < n_users && sc > tq[el & 3]) {
          const int pos = atomicAdd(&sCnt[ul], 1);
          sVal[ul * cap + pos] = sc;
          sIdx[ul * cap + pos] = item + item_base;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    wave_sync();
    for (int b0 = 0; b0 < UPW; b0 += 64) {
      const int ulq = uw + b0 + lane;
      unsigned long long over = __ballot(b0 + lane < UPW && sCnt[ulq] > settle_at);
      if constexpr (GBUF) {
        // (the appends are this wave's own stores: complete, and not served from a stale line of the vector cache)
        if (over) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
      }
      while (over) {
        const int ul = uw + b0 + __builtin_ctzll(over);
        over &= over - 1;
        if constexpr (GBUF)
          topk_settle_user_g(sVal + (size_t)ul * cap, sIdx + (size_t)ul * cap, sWrkV + wv * cap, sWrkI + wv * cap, sTmpV + wv * cap,
                             sTmpI + wv * cap, sCnt + ul, sThr + ul, sNeed + ul, topk, nr_ptr, nr_idx, u0 + ul, excl, n_excl, lane);
        else
          topk_settle_user(sVal + ul * cap, sIdx + ul * cap, sTmpV + wv * cap, sTmpI + wv * cap, sCnt + ul, sThr + ul, sNeed + ul,
                           topk, nr_ptr, nr_idx, u0 + ul, excl, n_excl, lane);
      }
    }
    store_tile(tl + 2);   // into the buffer of tile tl: its last reader finished before the previous barrier
    __syncthreads();
  };
  for (int tl = 0; tl < n_tiles; tl += 2) {
    body(tl, acc[0], acc[1]);
    if (tl + 1 < n_tiles) body(tl + 1, acc[1], acc[0]);
  }
  if constexpr (GBUF) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
  for (int ul = uw; ul < uw + UPW; ul++) {
    const int u = u0 + ul;
    if (u >= n_users) break;
    if constexpr (GBUF) {
      // always through the work area: the last arrivals are verified / reduced there, and the emission reads LDS
      float* wv_ = sWrkV + wv * cap;
      int* wi_ = sWrkI + wv * cap;
      const int n0 = sCnt[ul];
      wave_sync();
      for (int c = lane; c < n0; c += 64) {
        wv_[c] = sVal[(size_t)ul * cap + c];
        wi_[c] = sIdx[(size_t)ul * cap + c];
      }
      wave_sync();
      if (n0 > sNeed[ul])
        topk_settle_user(wv_, wi_, sTmpV + wv * cap, sTmpI + wv * cap, sCnt + ul, sThr + ul, sNeed + ul, topk, nr_ptr, nr_idx, u, excl,
                         n_excl, lane);
      topk_emit_user(wv_, wi_, min(sCnt[ul], topk), topk, lane, glob_mean, res + (size_t)u * topk, scores_out + (size_t)u * topk);
      continue;
    }
    if (sCnt[ul] > sNeed[ul])   // arrivals that no reduction has looked at yet (the buffer never filled up again)
      topk_settle_user(sVal + ul * cap, sIdx + ul * cap, sTmpV + wv * cap, sTmpI + wv * cap, sCnt + ul, sThr + ul, sNeed + ul, topk,
                       nr_ptr, nr_idx, u, excl, n_excl, lane);
    topk_emit_user(sVal + ul * cap, sIdx + ul * cap, min(sCnt[ul], topk), topk, lane, glob_mean, res + (size_t)u * topk,
                   scores_out + (size_t)u * topk);
  }
  __syncthreads();   // (the next block of this workgroup starts from the LDS areas again)
  }
}


// ---- few users, many items: the items split over the workgroups, the slices' lists merged (end of round 4) --------------------
// A call for a handful of users (the serving case) used to be ONE workgroup walking every item: 4.4 ms for one user over 100 k
// items, 33 ms for 1000 users, on 1..4 of 256 CUs.  Split launch: grid = (user blocks, item slices), every slice produces its own
// top k (exclusion lists applied, global item ids); one wave per user merges the S sorted lists.  The reference's result depends
// on the order of arrival only through candidates AT the k-th score (its heap replaces on strict >): the merged list is the
// sequential one whenever membership at that score is unambiguous -- no candidate left over at the k-th score, and no slice whose
// own k-th entry sits at it (that slice may have dropped equals).  Otherwise the user is flagged and the unsplit kernel, launched
// behind the merge with the flags, recomputes the workgroups that hold a flagged user (exact ties are degenerate inputs:
// tests/test_top_product.py has them).
__global__ __launch_bounds__(64) void top_merge_kernel(const float* __restrict__ sl_scores, const int32_t* __restrict__ sl_idx,
                                                       int n_slices, int n_users, int topk, float glob_mean,
                                                       int32_t* __restrict__ res, float* __restrict__ scores_out,
                                                       int* __restrict__ user_flags) {
  const int u = blockIdx.x, lane = threadIdx.x;
  const size_t ustride = (size_t)n_users * topk;
  const float* ls = sl_scores + (size_t)lane * ustride + (size_t)u * topk;
  const int32_t* li = sl_idx + (size_t)lane * ustride + (size_t)u * topk;
  const bool mine = lane < n_slices;
  int pos = 0;
  auto head = [&](float& hv, int& hi) {   // this slice's best unused entry (-inf: none)
    hv = -INFINITY;
    hi = INT32_MIN;
    if (mine && pos < topk) {
      const int ix = li[pos];
      if (ix != INT32_MIN) {
        hv = ls[pos];
        hi = ix;
      }
    }
  };
  float tau = -INFINITY;
  for (int out = 0; out < topk; out++) {
    float hv;
    int hi;
    head(hv, hi);
    float best = hv;
    for (int o = 32; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o));
    int bi = (hv == best && hi != INT32_MIN) ? hi : INT32_MIN;   // equal scores: the larger index first
    for (int o = 32; o > 0; o >>= 1) bi = max(bi, __shfl_xor(bi, o));
    if (bi == INT32_MIN) {   // fewer admissible items than k
      if (lane == 0) {
        res[(size_t)u * topk + out] = INT32_MIN;
        scores_out[(size_t)u * topk + out] = __int_as_float(0x7fc00000);
      }
      tau = -INFINITY;
    } else {
      if (lane == 0) {
        res[(size_t)u * topk + out] = bi;
        scores_out[(size_t)u * topk + out] = best + glob_mean;
      }
      if (hi == bi && hv == best) pos++;
      tau = best;
    }
  }
  // ambiguity at the k-th score
  float hv;
  int hi;
  head(hv, hi);
  bool amb = false;
  if (tau > -INFINITY) {
    amb = hv == tau;                                                      // a candidate left over at tau
    if (mine && li[topk - 1] != INT32_MIN && ls[topk - 1] == tau) amb = true;   // a full slice list that ends at tau
  }
  if (__any(amb) && lane == 0) user_flags[u] = 1;
}

}  // namespace

namespace {
hipError_t launch_top_product_geo(const float* U, const float* V, int n_users, int n_items, int k_rank, int topk,
                                  const int32_t* nr_ptr, const int32_t* nr_idx, const int32_t* excl, int n_excl,
                                  float glob_mean, int32_t* res, float* scores, hipStream_t s, int n_slices, int slice_items,
                                  const int* user_flags, float* gbuf = nullptr);
}

// ---- `$predict` that orders like the reference: candidates re-scored in double ----
// find_top_product casts both factor matrices to double before the product (R/utils.R:35-36) and top_product takes arma::mat
// (src/matrix_top_product.cpp:20): fp32 scores can swap items whose scores differ by ~1e-7 relative.  The fp32 matrix-core pass
// therefore only NOMINATES: it keeps the kc = k + extra best items of every user; one wave per user then recomputes those kc
// scores in double (from the double factors where the model holds them, else from the fp32 ones widened), and replays the
// reference's heap over the candidates in ascending item order.  That replay is the reference's result as long as every
// item whose double score reaches the k-th best is among the candidates: the final heap depends only on those items (items
// below the k-th score are evicted before any of them and never block one).  Equal scores: the larger index first; more
// candidates AT the k-th score than places: the survivors are the ones the heap keeps -- an arrival at the bound does not
// enter a full heap, a better arrival evicts the bound's smallest index.
// amb_out (pass 1, nullable): set for a user whose candidate list may not hold every item at the k-th score -- the list is full,
// its last entry sits AT that score, better items exist and the bound has more candidates than places: items of the bound
// that the larger heap of the fp32 pass evicted would have changed which of them the reference's heap keeps.  Such users are
// recomputed by the fp32 kernel with the reference's heap at capacity k (launch_top_product_f64) and only sorted here
// (pass 2: only_flagged = the same array, kc == topk).
template <class TF>
__global__ __launch_bounds__(256) void top_rescore_kernel(const TF* __restrict__ U, const TF* __restrict__ V, int n_users,
                                                          int rank, int kc, int topk, const int32_t* __restrict__ cand,
                                                          double glob_mean, int32_t* __restrict__ res,
                                                          double* __restrict__ scores, int* __restrict__ amb_out,
                                                          const int* __restrict__ only_flagged) {
  __shared__ double sU[4][256];
  __shared__ double sS[4][kTopMaxK];      // candidate scores
  __shared__ int sI[4][kTopMaxK];         // candidate items (0-based; -1 = none)
  __shared__ int sO[4][kTopMaxK];         // S (score >= v_k) in ascending item order: candidate slots
  __shared__ int sQ[4][kTopMaxK];         // replay: the bound's candidates inside the heap, oldest first
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int u = blockIdx.x * 4 + wv;
  if (u >= n_users) return;
  if (only_flagged && !only_flagged[u]) return;
  double* uu = sU[wv];
  double* cs = sS[wv];
  int* ci = sI[wv];
  for (int r = lane; r < rank; r += 64) uu[r] = (double)U[(size_t)u * rank + r];
  for (int c = lane; c < kc; c += 64) {
    const int ix = cand[(size_t)u * kc + c];
    ci[c] = ix == INT32_MIN ? -1 : ix - 1;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // scores: the wave walks the candidates, lane l takes the elements l, l + 64, ... of the item's vector
  for (int c = 0; c < kc; c++) {
    const int it = ci[c];
    double acc = 0.0;
    if (it >= 0)
      for (int r = lane; r < rank; r += 64) acc = fma(uu[r], (double)V[(size_t)it * rank + r], acc);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) cs[c] = acc;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // position of every candidate in the order (score descending, item descending); n_valid of them exist
  int n_valid = 0;
  for (int c = 0; c < kc; c++) n_valid += ci[c] >= 0 ? 1 : 0;
  const int kk = min(topk, n_valid);
  auto before = [&](const int a, const int b) {   // a ahead of b
    return cs[a] > cs[b] || (cs[a] == cs[b] && ci[a] > ci[b]);
  };
  constexpr int PER = kTopMaxK / 64;
  int pos[PER];
#pragma unroll
  for (int e = 0; e < PER; e++) {
    const int c = lane + 64 * e;
    int p = -1;
    if (c < kc && ci[c] >= 0) {
      p = 0;
      for (int o = 0; o < kc; o++) p += (ci[o] >= 0 && o != c && before(o, c)) ? 1 : 0;
    }
    pos[e] = p;
  }
  // the k-th best score, how many candidates beat it (g) and how many sit at it (nt)
  double vk = 0.0;
  int g = 0, nt = 0;
  if (kk > 0) {
    int who = -1;
#pragma unroll
    for (int e = 0; e < PER; e++)
      if (pos[e] == kk - 1) who = lane + 64 * e;
    const unsigned long long m = __ballot(who >= 0);
    const int src = __ffsll((long long)m) - 1;
    who = __shfl(who, src);
    vk = cs[who];
    double vmin = vk;
    for (int c = 0; c < kc; c++) {
      g += (ci[c] >= 0 && cs[c] > vk) ? 1 : 0;
      nt += (ci[c] >= 0 && cs[c] == vk) ? 1 : 0;
      if (ci[c] >= 0) vmin = fmin(vmin, cs[c]);
    }
    if (amb_out && lane == 0 && n_valid == kc && kc > topk && vmin == vk && g > 0 && g + nt > kk) amb_out[u] = 1;
  }
  int32_t* ru = res + (size_t)u * topk;
  double* su = scores + (size_t)u * topk;
  if (g + nt <= kk) {
    // no more candidates at the bound than places: the first kk positions are the answer
#pragma unroll
    for (int e = 0; e < PER; e++) {
      const int c = lane + 64 * e;
      if (pos[e] >= 0 && pos[e] < kk) {
        ru[pos[e]] = ci[c] + 1;
        su[pos[e]] = cs[c] + glob_mean;
      }
    }
  } else {
    // replay of the heap over S = {score >= v_k} in ascending item order (rare: exact ties at the bound)
    int* so = sO[wv];
    int* q = sQ[wv];
#pragma unroll
    for (int e = 0; e < PER; e++) {
      const int c = lane + 64 * e;
      if (c < kc && ci[c] >= 0 && cs[c] >= vk) {
        int r = 0;
        for (int o = 0; o < kc; o++) r += (ci[o] >= 0 && cs[o] >= vk && ci[o] < ci[c]) ? 1 : 0;
        so[r] = c;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int head = 0, tail = 0;
    if (lane == 0) {
      int cnt = 0;
      for (int e = 0; e < g + nt; e++) {
        const int c = so[e];
        if (cs[c] > vk) {
          if (cnt < kk) cnt++;
          else head++;               // evicts the smallest (score, item) pair: the bound's oldest = smallest index
        } else if (cnt < kk) {
          q[tail++] = c;
          cnt++;
        }
      }
    }
    head = __shfl(head, 0);
    tail = __shfl(tail, 0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int e = 0; e < PER; e++) {
      const int c = lane + 64 * e;
      if (pos[e] >= 0 && cs[c] > vk) {   // better than the bound: positions 0 .. g - 1 as they stand
        ru[pos[e]] = ci[c] + 1;
        su[pos[e]] = cs[c] + glob_mean;
      }
    }
    // the survivors at the bound, larger index first: q[head .. tail) is ascending in item
    for (int e = head + lane; e < tail; e += 64) {
      const int c = q[e];
      const int at = g + (tail - 1 - e);
      ru[at] = ci[c] + 1;
      su[at] = cs[c] + glob_mean;
    }
  }
  for (int c = kk + lane; c < topk; c += 64) {   // fewer admissible items than k: NA_integer_ / NA_real_
    ru[c] = INT32_MIN;
    su[c] = __longlong_as_double(0x7ff8000000000000ll);
  }
}

// The whole `$predict` that orders like the reference.  U32 / V32: the fp32 factors (the nominating pass); U64 / V64: the double
// ones (nullable as a pair: the fp32 factors widened).  scratch: n_users * (kc + topk) ints + as many floats + n_users ints
// (top_product_f64_scratch_words).  Steps: (1) the fp32 kernel keeps kc = k + extra candidates per user; (2) one wave per user
// recomputes their scores in double and replays the reference's heap (top_rescore_kernel), flagging the users whose candidate
// list may be missing items at the k-th score; (3) the workgroups of the fp32 kernel that hold a flagged user run again at
// capacity k -- the reference's own heap, exact ties included -- and (4) those users' k items are re-scored and sorted.  No host
// round trip: without flagged users (3) and (4) are launches whose workgroups return at once.
size_t top_product_f64_scratch_words(int n_users, int kc, int topk) {
  return (size_t)n_users * ((size_t)2 * kc + 2 * topk + 1) + 64;
}
template <class TF>
hipError_t launch_top_product_f64_t(const float* U32, const float* V32, const TF* U, const TF* V, int n_users, int n_items,
                                    int rank, int topk, int kc, const int32_t* nr_ptr, const int32_t* nr_idx, const int32_t* excl,
                                    int n_excl, double glob_mean, int32_t* res, double* scores, hipStream_t s, float* scratch,
                                    float* split_scratch) {
  if (n_users <= 0) return hipSuccess;
  if (kc < topk || kc > kTopMaxK || topk < 1 || rank < 1 || rank > 256) return hipErrorInvalidValue;
  int32_t* cand = reinterpret_cast<int32_t*>(scratch);
  float* cand_sc = scratch + (size_t)n_users * kc;
  int32_t* res2 = reinterpret_cast<int32_t*>(cand_sc + (size_t)n_users * kc);
  float* sc2 = reinterpret_cast<float*>(res2 + (size_t)n_users * topk);
  int* flags = reinterpret_cast<int*>(sc2 + (size_t)n_users * topk);
  hipError_t err;
  if ((err = launch_top_product(U32, V32, n_users, n_items, rank, kc, nr_ptr, nr_idx, excl, n_excl, 0.f, cand, cand_sc, s,
                                split_scratch)) != hipSuccess)
    return err;
  if ((err = hipMemsetAsync(flags, 0, (size_t)n_users * sizeof(int), s)) != hipSuccess) return err;
  const dim3 grid((n_users + 3) / 4);
  hipLaunchKernelGGL(top_rescore_kernel<TF>, grid, dim3(256), 0, s, U, V, n_users, rank, kc, topk, cand, glob_mean, res, scores,
                     flags, static_cast<const int*>(nullptr));
  if ((err = hipGetLastError()) != hipSuccess) return err;
  if (kc > topk) {
    // (the tile-sharing kernel with its block loop and the global buffers where the scratch was sized for them -- an unsliced call
    //  for more than 128 users --, else the geometry the LDS allows)
    float* rerun_gbuf = (split_scratch && top_product_scratch_entries(n_users, n_items, kc) == 0 && top_product_wants_gbuf(n_users, rank, topk))
                            ? split_scratch : nullptr;
    if ((err = launch_top_product_geo(U32, V32, n_users, n_items, rank, topk, nr_ptr, nr_idx, excl, n_excl, 0.f, res2, sc2, s, 1, 0,
                                      flags, rerun_gbuf)) != hipSuccess)
      return err;
    hipLaunchKernelGGL(top_rescore_kernel<TF>, grid, dim3(256), 0, s, U, V, n_users, rank, topk, topk, res2, glob_mean, res, scores,
                       static_cast<int*>(nullptr), flags);
    if ((err = hipGetLastError()) != hipSuccess) return err;
  }
  return hipSuccess;
}
hipError_t launch_top_product_f64(const float* U32, const float* V32, const double* U64, const double* V64, int n_users,
                                  int n_items, int rank, int topk, int kc, const int32_t* nr_ptr, const int32_t* nr_idx,
                                  const int32_t* excl, int n_excl, double glob_mean, int32_t* res, double* scores, hipStream_t s,
                                  float* scratch, float* split_scratch) {
  if (U64 && V64)
    return launch_top_product_f64_t<double>(U32, V32, U64, V64, n_users, n_items, rank, topk, kc, nr_ptr, nr_idx, excl, n_excl,
                                            glob_mean, res, scores, s, scratch, split_scratch);
  return launch_top_product_f64_t<float>(U32, V32, U32, V32, n_users, n_items, rank, topk, kc, nr_ptr, nr_idx, excl, n_excl,
                                         glob_mean, res, scores, s, scratch, split_scratch);
}

// how a call is split: slices of `slice_items` items (a multiple of 32), 1 = not at all
int top_product_slices(int n_users, int n_items, int topk, int* slice_items) {
  *slice_items = 0;
  // (users per workgroup of the geometry launch_top_product_geo picks for this many users; a smaller one -- when the candidate
  //  buffers of a large k do not fit -- only means more workgroups than planned)
  const int upw = n_users > 128 ? 256 : (n_users > 64 ? 128 : (n_users > 32 ? 64 : 32));
  const int ublocks = (n_users + upw - 1) / upw;
  const int n_tiles = (n_items + 31) / 32;
  if (n_users <= 0 || ublocks >= 128 || n_tiles < 128) return 1;   // enough workgroups already, or nothing to split
  int S = std::min(64, std::min((512 + ublocks - 1) / ublocks, n_tiles / 32));
  if ((size_t)S * n_users * topk > ((size_t)8 << 20)) S = (int)(((size_t)8 << 20) / ((size_t)n_users * topk));
  if (S < 2) return 1;
  const int st = (n_tiles + S - 1) / S;
  *slice_items = 32 * st;
  return (n_tiles + st - 1) / st;
}
size_t top_product_scratch_entries(int n_users, int n_items, int topk) {
  int si = 0;
  const int S = top_product_slices(n_users, n_items, topk, &si);
  return S > 1 ? (size_t)S * n_users * topk : 0;
}
// GBUF: does a call of this shape keep its candidate buffers in the global scratch?  Every call for more than 128 users at a rank
// up to 128 (at 129..256 a wave holds one user block): measured faster than the LDS buffers at EVERY k -- top-100 29 -> 86
// TFLOP/s (the LDS took 32 users per workgroup there), top-10 80 -> 96, top-1 89 -> 98 (profiles/r06/r6topk_*).
bool top_product_wants_gbuf(int n_users, int k_rank, int topk) {
  if (n_users <= 128 || k_rank > 128 || topk < 1 || topk > kTopMaxK) return false;
#ifdef RSP_TOPK_NO_GBUF   // dev builds: the round-5 geometries
  return false;
#endif
  return padded_rank(k_rank) != 0;
}
// floats of scratch launch_top_product wants for this call (0 = none): the slices' lists of a call for few users, or the
// global candidate buffers of a call for many users at a large k (at most 512 workgroup slots: longer calls go in chunks)
size_t top_product_scratch_floats(int n_users, int n_items, int k_rank, int topk) {
  const size_t ent = top_product_scratch_entries(n_users, n_items, topk);
  if (ent > 0) return 2 * ent + (size_t)n_users + 16;
  if (!top_product_wants_gbuf(n_users, k_rank, topk)) return 0;
  const size_t blocks = ((size_t)std::min(n_users, kTopGbufChunk) + 255) / 256;
  return blocks * 2 * 256 * (size_t)top_gcap(topk) + 16;
}

// scratch (nullable): top_product_scratch_entries floats + as many ints + n_users ints
hipError_t launch_top_product(const float* U, const float* V, int n_users, int n_items, int k_rank, int topk,
                              const int32_t* nr_ptr, const int32_t* nr_idx, const int32_t* excl, int n_excl,
                              float glob_mean, int32_t* res, float* scores, hipStream_t s, float* scratch) {
  int slice_items = 0;
  const int S = scratch ? top_product_slices(n_users, n_items, topk, &slice_items) : 1;
  if (S <= 1) {
    if (scratch && top_product_wants_gbuf(n_users, k_rank, topk)) {
      // the candidate buffers in the scratch: chunks of users, one after the other on the stream, share its 512 slots
      for (int c0 = 0; c0 < n_users; c0 += kTopGbufChunk) {
        const int nu = std::min(kTopGbufChunk, n_users - c0);
        const hipError_t e = launch_top_product_geo(U + (size_t)c0 * k_rank, V, nu, n_items, k_rank, topk, nr_ptr ? nr_ptr + c0 : nullptr,
                                                    nr_idx, excl, n_excl, glob_mean, res + (size_t)c0 * topk, scores + (size_t)c0 * topk, s,
                                                    1, 0, nullptr, scratch);
        if (e != hipSuccess) return e;
      }
      return hipSuccess;
    }
    return launch_top_product_geo(U, V, n_users, n_items, k_rank, topk, nr_ptr, nr_idx, excl, n_excl, glob_mean, res, scores, s, 1,
                                  0, nullptr);
  }
  const size_t ent = (size_t)S * n_users * topk;
  float* sl_scores = scratch;
  int32_t* sl_idx = reinterpret_cast<int32_t*>(scratch + ent);
  int* flags = reinterpret_cast<int*>(sl_idx + ent);
  hipError_t err;
  if ((err = hipMemsetAsync(flags, 0, (size_t)n_users * sizeof(int), s)) != hipSuccess) return err;
  if ((err = launch_top_product_geo(U, V, n_users, n_items, k_rank, topk, nr_ptr, nr_idx, excl, n_excl, 0.f, sl_idx, sl_scores, s, S,
                                    slice_items, nullptr)) != hipSuccess)
    return err;
  hipLaunchKernelGGL(top_merge_kernel, dim3(n_users), dim3(64), 0, s, sl_scores, sl_idx, S, n_users, topk, glob_mean, res, scores,
                     flags);
  if ((err = hipGetLastError()) != hipSuccess) return err;
  return launch_top_product_geo(U, V, n_users, n_items, k_rank, topk, nr_ptr, nr_idx, excl, n_excl, glob_mean, res, scores, s, 1, 0,
                                flags);
}

namespace {
hipError_t launch_top_product_geo(const float* U, const float* V, int n_users, int n_items, int k_rank, int topk,
                                  const int32_t* nr_ptr, const int32_t* nr_idx, const int32_t* excl, int n_excl,
                                  float glob_mean, int32_t* res, float* scores, hipStream_t s, int n_slices, int slice_items,
                                  const int* user_flags, float* gbuf) {
  const int KP = (k_rank > 128 && k_rank <= 256) ? 256 : padded_rank(k_rank);   // (ranks 129..256: one user block per wave)
  if (!KP || topk < 1 || topk > kTopMaxK) return hipErrorInvalidValue;
  if (n_users <= 0) return hipSuccess;
  const bool vec = k_rank % 4 == 0 && (reinterpret_cast<uintptr_t>(V) & 15) == 0;
  constexpr size_t kLdsMax = 156 * 1024;
  hipError_t err;
#define RSP_TOPK_GO(KERN, LDS, USERS, THREADS)                                                               \
  {                                                                                                          \
    auto kern = KERN;                                                                                        \
    const size_t lds = LDS;                                                                                  \
    if ((err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   (int)lds)) != hipSuccess)                                                 \
      return err;                                                                                            \
    const int grid = (n_users + (USERS) - 1) / (USERS);                                                      \
    hipLaunchKernelGGL(kern, dim3(grid, n_slices), dim3(THREADS), lds, s, U, V, n_users, n_items, k_rank, topk, nr_ptr, \
                       nr_idx, excl, n_excl, glob_mean, res, scores, slice_items, user_flags);               \
    return hipGetLastError();                                                                                \
  }
  // (the two-tile kernel takes the global candidate scratch as well)
#define RSP_TOPK_GO_P(KERN, LDS, USERS, THREADS, GB)                                                         \
  {                                                                                                          \
    auto kern = KERN;                                                                                        \
    const size_t lds = LDS;                                                                                  \
    if ((err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   (int)lds)) != hipSuccess)                                                 \
      return err;                                                                                            \
    const int grid = std::min((n_users + (USERS) - 1) / (USERS), user_flags ? 512 : 0x7fffffff);  /* (see the kernel's block loop) */ \
    hipLaunchKernelGGL(kern, dim3(grid, n_slices), dim3(THREADS), lds, s, U, V, n_users, n_items, k_rank, topk, nr_ptr, \
                       nr_idx, excl, n_excl, glob_mean, res, scores, slice_items, user_flags, GB);           \
    return hipGetLastError();                                                                                \
  }
  // GBUF: the launcher hands a scratch when the call is unsliced; taken where the 256-user buffers do not fit the LDS
  // (a flagged launch has at most 512 workgroups = scratch slots, whatever the number of users)
  const bool gbuf_now = gbuf && n_slices == 1 && n_users > 128 && (n_users <= kTopGbufChunk || user_flags) && top_product_wants_gbuf(n_users, k_rank, topk);
  // Geometry, best first: the four waves share the item tile and own 64 / 32 users each (256 / 128 users per workgroup:
  // every item vector is read once per that many users) -- with two tiles resident and the epilogue hidden behind the next
  // tile's matrix instructions where the LDS allows it; then one tile per wave against 64 / 32 users; then the same on
  // two waves (half the tiles and half the per-round candidates in LDS: top-k up to 256 at rank 128).  What decides is
  // whether the users' candidate buffers fit the LDS next to the tiles, and that there are users enough for the block.
#define RSP_TOPK(KPV)                                                                                        \
  if (KP == KPV) {                                                                                           \
    if (gbuf_now) {                                                                                          \
      if (vec) RSP_TOPK_GO_P((top_product_pipe_kernel<KPV, 2, true, true>), (TopPipeSmem<KPV, 2>::gbytes(topk)), 256, 256, gbuf) \
      else RSP_TOPK_GO_P((top_product_pipe_kernel<KPV, 2, false, true>), (TopPipeSmem<KPV, 2>::gbytes(topk)), 256, 256, gbuf)     \
    }                                                                                                        \
    if (TopPipeSmem<KPV, 2>::bytes(topk) <= kLdsMax && n_users > 128) {                                      \
      if (vec) RSP_TOPK_GO_P((top_product_pipe_kernel<KPV, 2, true>), (TopPipeSmem<KPV, 2>::bytes(topk)), 256, 256, nullptr) \
      else RSP_TOPK_GO_P((top_product_pipe_kernel<KPV, 2, false>), (TopPipeSmem<KPV, 2>::bytes(topk)), 256, 256, nullptr)     \
    }                                                                                                        \
    if (TopPipeSmem<KPV, 1>::bytes(topk) <= kLdsMax && n_users > 64) {                                       \
      if (vec) RSP_TOPK_GO_P((top_product_pipe_kernel<KPV, 1, true>), (TopPipeSmem<KPV, 1>::bytes(topk)), 128, 256, nullptr) \
      else RSP_TOPK_GO_P((top_product_pipe_kernel<KPV, 1, false>), (TopPipeSmem<KPV, 1>::bytes(topk)), 128, 256, nullptr)     \
    }                                                                                                        \
    if (TopSharedSmem<KPV, 2>::bytes(topk) <= kLdsMax && n_users > 128) {                                    \
      if (vec) RSP_TOPK_GO((top_product_shared_kernel<KPV, 2, true>), (TopSharedSmem<KPV, 2>::bytes(topk)), 256, 256) \
      else RSP_TOPK_GO((top_product_shared_kernel<KPV, 2, false>), (TopSharedSmem<KPV, 2>::bytes(topk)), 256, 256)     \
    }                                                                                                        \
    if (TopSharedSmem<KPV, 1>::bytes(topk) <= kLdsMax && n_users > 64) {                                     \
      if (vec) RSP_TOPK_GO((top_product_shared_kernel<KPV, 1, true>), (TopSharedSmem<KPV, 1>::bytes(topk)), 128, 256) \
      else RSP_TOPK_GO((top_product_shared_kernel<KPV, 1, false>), (TopSharedSmem<KPV, 1>::bytes(topk)), 128, 256)     \
    }                                                                                                        \
    if (TopSmem<KPV, 2, 4>::bytes(topk) <= kLdsMax && n_users > 32) {                                        \
      if (vec) RSP_TOPK_GO((top_product_kernel<KPV, 2, true, 4>), (TopSmem<KPV, 2, 4>::bytes(topk)), 64, 256) \
      else RSP_TOPK_GO((top_product_kernel<KPV, 2, false, 4>), (TopSmem<KPV, 2, 4>::bytes(topk)), 64, 256)    \
    }                                                                                                        \
    if (TopSmem<KPV, 1, 4>::bytes(topk) <= kLdsMax) {                                                        \
      if (vec) RSP_TOPK_GO((top_product_kernel<KPV, 1, true, 4>), (TopSmem<KPV, 1, 4>::bytes(topk)), 32, 256) \
      else RSP_TOPK_GO((top_product_kernel<KPV, 1, false, 4>), (TopSmem<KPV, 1, 4>::bytes(topk)), 32, 256)    \
    }                                                                                                        \
    if (vec) RSP_TOPK_GO((top_product_kernel<KPV, 1, true, 2>), (TopSmem<KPV, 1, 2>::bytes(topk)), 32, 128)  \
    else RSP_TOPK_GO((top_product_kernel<KPV, 1, false, 2>), (TopSmem<KPV, 1, 2>::bytes(topk)), 32, 128)     \
  }
  RSP_TOPK(32)
  RSP_TOPK(64)
  RSP_TOPK(128)
  if (KP == 256) {   // the user block of a wave is 128 registers at this rank: one block per wave in every geometry
    if (TopPipeSmem<256, 1>::bytes(topk) <= kLdsMax && n_users > 64) {
      if (vec) RSP_TOPK_GO_P((top_product_pipe_kernel<256, 1, true>), (TopPipeSmem<256, 1>::bytes(topk)), 128, 256, nullptr)
      else RSP_TOPK_GO_P((top_product_pipe_kernel<256, 1, false>), (TopPipeSmem<256, 1>::bytes(topk)), 128, 256, nullptr)
    }
    if (TopSharedSmem<256, 1>::bytes(topk) <= kLdsMax && n_users > 64) {
      if (vec) RSP_TOPK_GO((top_product_shared_kernel<256, 1, true>), (TopSharedSmem<256, 1>::bytes(topk)), 128, 256)
      else RSP_TOPK_GO((top_product_shared_kernel<256, 1, false>), (TopSharedSmem<256, 1>::bytes(topk)), 128, 256)
    }
    if (TopSmem<256, 1, 4>::bytes(topk) <= kLdsMax) {
      if (vec) RSP_TOPK_GO((top_product_kernel<256, 1, true, 4>), (TopSmem<256, 1, 4>::bytes(topk)), 32, 256)
      else RSP_TOPK_GO((top_product_kernel<256, 1, false, 4>), (TopSmem<256, 1, 4>::bytes(topk)), 32, 256)
    }
    if (vec) RSP_TOPK_GO((top_product_kernel<256, 1, true, 2>), (TopSmem<256, 1, 2>::bytes(topk)), 32, 128)
    else RSP_TOPK_GO((top_product_kernel<256, 1, false, 2>), (TopSmem<256, 1, 2>::bytes(topk)), 32, 128)
  }
#undef RSP_TOPK
#undef RSP_TOPK_GO
#undef RSP_TOPK_GO_P
  return hipErrorInvalidValue;
}
}  // namespace

}  // namespace rsparse_hip
