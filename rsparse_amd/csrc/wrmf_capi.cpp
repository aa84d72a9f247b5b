// C ABI of librsparse_wrmf_hip.so (declared in include/rsparse_wrmf_hip.h).
//
// Host-side responsibilities only: argument validation with the reference's error behaviour turned
// into status codes (the reference raises R errors through END_RCPP, src/RcppExports.cpp:374,392),
// device buffers, the row-length schedule, and the bookkeeping around the kernels in
// wrmf_kernels.hip.  No torch, no R, no Armadillo types anywhere.
#include "../../include/rsparse_wrmf_hip.h"

#include <cxxabi.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <queue>
#include <string>
#include <vector>

#include "wrmf_internal.h"

using namespace rsparse_hip;

struct rsparse_hip_csc {
  DevCSC d;
  int device = 0;
};

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
int hip_fail(hipError_t e, const char* what) {
  return fail(RSPARSE_HIP_ERR_RUNTIME, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr)                                       \
  do {                                                      \
    hipError_t _e = (expr);                                 \
    if (_e != hipSuccess) return hip_fail(_e, #expr);       \
  } while (0)

// Grow-only per-process scratch (single host thread drives the library, like the reference's
// single R thread; not re-entrant across streams).
struct Workspace {
  float* gram = nullptr;
  size_t gram_floats = 0;
  double* partials = nullptr;
  size_t partial_slots = 0;
  double* scalars = nullptr;  // [0] loss rows, [1] sumsq, [2..] spare
  int* fails = nullptr;
  unsigned* ne_stats = nullptr;   // normal-equation kernel: max |x|, max c, any c < 1 (launch_ne_stats)
  float* lr_M = nullptr;
  float* ne_seg_scratch = nullptr;
  int* ne_seg_flags = nullptr;
  size_t ne_seg_slots = 0;
  float* zero_row = nullptr;  // 256 zero floats (padding slots of the CG gathers)
  float* tscr = nullptr;   // streamed CG rows: per-non-zero dot products of every sweep
  size_t tscr_floats = 0;
  float* bias_buf = nullptr;   // explicit + biases: X', Y', shifted ratings
  size_t bias_floats = 0;
  float* gb_r0 = nullptr;      // global bias + CG: per long row, base - g X_nnz (c - 1)  (launch_gb_row_terms)
  size_t gb_r0_floats = 0;
  int32_t* gb_slot = nullptr;  // ... and the slot of every row in it
  size_t gb_slot_ints = 0;
  float* wide_m2 = nullptr;    // ranks 129..256 (wrmf_wide.hip): NNLS squared systems / the general solver's matrices
  size_t wide_m2_floats = 0;
  float* wide_lu = nullptr;
  size_t wide_lu_floats = 0;
  int device = -1;

  int ensure_device() {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != device) {  // buffers belong to the device they were allocated on
      release();
      device = dev;
    }
    if (!scalars) {
      HIP_TRY(hipMalloc(&scalars, 16 * sizeof(double)));
      HIP_TRY(hipMemset(scalars, 0, 16 * sizeof(double)));
    }
    if (!fails) {   // see launch_fail_roll
      HIP_TRY(hipMalloc(&fails, (size_t)(4 + kFailCap) * sizeof(int)));
      HIP_TRY(hipMemset(fails, 0, 4 * sizeof(int)));
    }
    if (!ne_stats) {
      HIP_TRY(hipMalloc(&ne_stats, 4 * sizeof(unsigned)));
      HIP_TRY(hipMemset(ne_stats, 0, 4 * sizeof(unsigned)));
    }
    if (!zero_row) {
      HIP_TRY(hipMalloc(&zero_row, 256 * sizeof(float)));
      HIP_TRY(hipMemset(zero_row, 0, 256 * sizeof(float)));
    }
    return RSPARSE_HIP_OK;
  }
  int ensure_lr() {   // M = L^-T and its transpose of the low-rank Cholesky path
    if (!lr_M) HIP_TRY(hipMalloc(&lr_M, (size_t)3 * 128 * 128 * sizeof(float)));
    return RSPARSE_HIP_OK;
  }
  int ensure_ne_seg(size_t slots) {   // split rows of the normal-equation kernel: partial accumulators + ready flags
    if (slots > ne_seg_slots) {
      if (ne_seg_scratch) (void)hipFree(ne_seg_scratch);
      if (ne_seg_flags) (void)hipFree(ne_seg_flags);
      ne_seg_scratch = nullptr; ne_seg_flags = nullptr; ne_seg_slots = 0;
      HIP_TRY(hipMalloc(&ne_seg_scratch, slots * (size_t)kNeSegFloats * sizeof(float)));
      HIP_TRY(hipMalloc(&ne_seg_flags, slots * sizeof(int)));
      ne_seg_slots = slots;
    }
    return RSPARSE_HIP_OK;
  }
  int ensure_gram(size_t floats) {
    if (floats > gram_floats) {
      if (gram) (void)hipFree(gram);
      gram = nullptr;
      gram_floats = 0;
      HIP_TRY(hipMalloc(&gram, floats * sizeof(float)));
      gram_floats = floats;
    }
    return RSPARSE_HIP_OK;
  }
  int ensure_partials(size_t slots) {
    if (slots < 1024) slots = 1024;
    if (slots > partial_slots) {
      if (partials) (void)hipFree(partials);
      partials = nullptr;
      partial_slots = 0;
      HIP_TRY(hipMalloc(&partials, (slots + kSumStageBlocks) * sizeof(double)));   // + the tail of the two-stage sum
      partial_slots = slots;
    }
    return RSPARSE_HIP_OK;
  }
  float* mf_G = nullptr;      // wrmf_chol_mf.hip at rank 65..127: XtX padded to 128 x 128
  int ensure_mf() {
    if (!mf_G) HIP_TRY(hipMalloc(&mf_G, (size_t)128 * 128 * sizeof(float)));
    return RSPARSE_HIP_OK;
  }
  float* pad_buf = nullptr;   // ranks that are not a multiple of 4: the padded copies of X, Y, XtX, rhs_init (run_half_iteration)
  size_t pad_floats = 0;
  int ensure_pad(size_t floats) {
    if (floats > pad_floats) {
      if (pad_buf) (void)hipFree(pad_buf);
      pad_buf = nullptr;
      pad_floats = 0;
      HIP_TRY(hipMalloc(&pad_buf, floats * sizeof(float)));
      pad_floats = floats;
    }
    return RSPARSE_HIP_OK;
  }
  int ensure_bias(size_t floats) {
    if (floats > bias_floats) {
      if (bias_buf) (void)hipFree(bias_buf);
      bias_buf = nullptr;
      bias_floats = 0;
      HIP_TRY(hipMalloc(&bias_buf, floats * sizeof(float)));
      bias_floats = floats;
    }
    return RSPARSE_HIP_OK;
  }
  int ensure_gb(size_t floats, size_t ints) {
    if (floats > gb_r0_floats) {
      if (gb_r0) (void)hipFree(gb_r0);
      gb_r0 = nullptr; gb_r0_floats = 0;
      HIP_TRY(hipMalloc(&gb_r0, floats * sizeof(float)));
      gb_r0_floats = floats;
    }
    if (ints > gb_slot_ints) {
      if (gb_slot) (void)hipFree(gb_slot);
      gb_slot = nullptr; gb_slot_ints = 0;
      HIP_TRY(hipMalloc(&gb_slot, ints * sizeof(int32_t)));
      gb_slot_ints = ints;
    }
    return RSPARSE_HIP_OK;
  }
  int ensure_wide(size_t m2, size_t lu) {
    if (m2 > wide_m2_floats) {
      if (wide_m2) (void)hipFree(wide_m2);
      wide_m2 = nullptr; wide_m2_floats = 0;
      HIP_TRY(hipMalloc(&wide_m2, m2 * sizeof(float)));
      wide_m2_floats = m2;
    }
    if (lu > wide_lu_floats) {
      if (wide_lu) (void)hipFree(wide_lu);
      wide_lu = nullptr; wide_lu_floats = 0;
      HIP_TRY(hipMalloc(&wide_lu, lu * sizeof(float)));
      wide_lu_floats = lu;
    }
    return RSPARSE_HIP_OK;
  }
  int ensure_tscr(size_t floats) {
    if (floats > tscr_floats) {
      if (tscr) (void)hipFree(tscr);
      tscr = nullptr;
      tscr_floats = 0;
      HIP_TRY(hipMalloc(&tscr, floats * sizeof(float)));
      tscr_floats = floats;
    }
    return RSPARSE_HIP_OK;
  }
  void release() {
    if (wide_m2) (void)hipFree(wide_m2);
    if (wide_lu) (void)hipFree(wide_lu);
    wide_m2 = wide_lu = nullptr; wide_m2_floats = wide_lu_floats = 0;
    if (gb_r0) (void)hipFree(gb_r0);
    if (gb_slot) (void)hipFree(gb_slot);
    gb_r0 = nullptr; gb_slot = nullptr; gb_r0_floats = 0; gb_slot_ints = 0;
    if (tscr) (void)hipFree(tscr);
    tscr = nullptr;
    tscr_floats = 0;
    if (bias_buf) (void)hipFree(bias_buf);
    bias_buf = nullptr;
    bias_floats = 0;
    if (gram) (void)hipFree(gram);
    if (partials) (void)hipFree(partials);
    if (scalars) (void)hipFree(scalars);
    if (fails) (void)hipFree(fails);
    if (ne_stats) (void)hipFree(ne_stats);
    ne_stats = nullptr;
    if (ne_seg_scratch) (void)hipFree(ne_seg_scratch);
    if (ne_seg_flags) (void)hipFree(ne_seg_flags);
    ne_seg_scratch = nullptr; ne_seg_flags = nullptr; ne_seg_slots = 0;
    if (lr_M) (void)hipFree(lr_M);
    lr_M = nullptr;
    if (mf_G) (void)hipFree(mf_G);
    mf_G = nullptr;
    if (pad_buf) (void)hipFree(pad_buf);
    pad_buf = nullptr; pad_floats = 0;
    if (zero_row) (void)hipFree(zero_row);
    gram = nullptr; partials = nullptr; scalars = nullptr; fails = nullptr; zero_row = nullptr;
    gram_floats = 0; partial_slots = 0;
  }
};
thread_local Workspace g_ws;   // per host thread: the multi-GPU context (wrmf_ctx.cpp) drives one device from one thread each

struct Profiler {
  bool on = false;
  bool have = false;
  int nseg = 0;
  hipEvent_t ev[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  const void* kern[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t* begin() {
    have = false;
    if (!on) return nullptr;
    for (auto& e : ev)
      if (!e && hipEventCreate(&e) != hipSuccess) return nullptr;
    for (auto& k : kern) k = nullptr;
    return ev;
  }
};
thread_local Profiler g_prof;

struct DevBuf {  // RAII for the stateless entry points
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
  template <class T> T* as() { return static_cast<T*>(p); }
};

// rows with more than kTileNnz non-zeros, longest first (counting sort on the host; one-off per matrix)
int build_schedule(DevCSC& d, const int32_t* host_col_ptrs) {
  const int n = d.n_cols;
  int max_len = 0;
  size_t n_long = 0;
  int64_t nnz_long = 0;
  int n_empty = 0;
  for (int i = 0; i < n; i++) {
    const int len = host_col_ptrs[i + 1] - host_col_ptrs[i];
    if (len < 0) return fail(RSPARSE_HIP_ERR_INVALID, "col_ptrs is not non-decreasing");
    max_len = std::max(max_len, len);
    if (len > kTileNnz) { n_long++; nnz_long += len; }
    if (len == 0) n_empty++;
  }
  d.max_len = max_len;
  d.nnz_long = nnz_long;
  d.n_empty = n_empty;
  d.short_max = kTileNnz;
  d.n_long = (int)n_long;
  d.long_rows = nullptr;
  if (!n_long) return RSPARSE_HIP_OK;
  // counting sort by length, descending; ties keep ascending row order (deterministic)
  std::vector<int64_t> start((size_t)max_len + 2, 0);
  for (int i = 0; i < n; i++) {
    const int len = host_col_ptrs[i + 1] - host_col_ptrs[i];
    if (len > kTileNnz) start[(size_t)(max_len - len) + 1]++;
  }
  for (size_t b = 1; b < start.size(); b++) start[b] += start[b - 1];
  std::vector<int32_t> order(n_long);
  for (int i = 0; i < n; i++) {
    const int len = host_col_ptrs[i + 1] - host_col_ptrs[i];
    if (len > kTileNnz) order[(size_t)start[(size_t)(max_len - len)]++] = i;
  }
  HIP_TRY(hipMalloc(&d.long_rows, n_long * sizeof(int32_t)));
  HIP_TRY(hipMemcpy(d.long_rows, order.data(), n_long * sizeof(int32_t), hipMemcpyHostToDevice));
  return RSPARSE_HIP_OK;
}

struct NeLists {
  int32_t* rows = nullptr; int32_t* ptr = nullptr; int wg = 0;
  int32_t* segs = nullptr; int nseg = 0; int entries = 0;
  int32_t* split_rows = nullptr; int32_t* split_ptr = nullptr; int nsplit = 0;
};

// Lists of the normal-equation launch over the n_prefix longest rows (order = every row, longest first); `fixed` = the
// per-row cost of the solve in 16-non-zero steps (CG: 12; the exact solve of solver == CHOLESKY: 72).
int build_ne_lists(const std::vector<int32_t>& order, const int32_t* host_col_ptrs, int n_prefix, int64_t fixed, NeLists& L,
                   bool fine = true) {
  if (n_prefix <= 0) return RSPARSE_HIP_OK;
  auto len_of = [&](int r) { return (int64_t)(host_col_ptrs[order[(size_t)r] + 1] - host_col_ptrs[order[(size_t)r]]); };
  {
  // Row lists of the normal-equation kernel: one workgroup per CU, rows dealt longest-processing-time first (the rows
  // arrive sorted by length, each goes to the least loaded workgroup; cost = the row's 16-non-zero steps + a fixed
  // per-row solve).  Static lists make the per-row loss slots and the summation order deterministic.
  int dev = 0, cus = 256;
  HIP_TRY(hipGetDevice(&dev));
  HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  // Many more lists than workgroup slots (eight rows or more per list, up to 256 lists per CU; two workgroups of the
  // rank-128 fp16 kernel are resident per CU): the
  // hardware hands the next list to whichever slot frees up.  With exactly one list per slot the launch ended 13 % after
  // its mean workgroup (round 3, in-kernel counters): of the two workgroups that share a CU's SIMDs the one dispatched
  // first wins the issue arbitration and runs 27 % faster -- every workgroup of index < 256 took 63.9 M ticks for its
  // list, every one of index >= 256 81.4 M for an equal list, the last 17 M of them alone on its CU.
  // (round 6, `fine` lists: the slots of the whole machine even when there are fewer rows than slots -- the giant rows that
  //  wrmf_cg_mf.hip leaves to this kernel are a few hundred: one workgroup per row left 40 % of the slots empty and every row as
  //  long as its wave could stream it, 4.4 GB in 4.7 ms; cut to the machine's share they are segments of >= 64 steps)
  // The CUT must not depend on the deal: the two list sets of a matrix (fine / one list per slot) share ONE segment table and
  // one list of split rows (build_q_schedule) -- cut by the fine rule only, a row was whole in the coarse lists and "split" in
  // the table, and the collecting launch overwrote its solution with the sum of two stale partials (ranks up to 96, fewer
  // than 512 long rows, a row of >= 2048 non-zeros: tests/test_bias.py caught it at the end of round 6).
  const int n_slots = 2 * std::max(cus, 1);   // what the share of the split rule refers to
  // (`fine` = false: one list per slot, for the kernels that are resident once per CU -- no such asymmetry there, and a
  //  workgroup start costs more: XtX tiles into LDS; many short lists cost them 1..7 %)
  int n_wg = fine ? std::max(n_slots, std::min(n_prefix / 8, 256 * std::max(cus, 1))) : n_slots;   // (at most one per item: below)
  // Items of the deal: whole rows, and SEGMENTS of the rows that are too long to balance (the 5e5-non-zero item of the
  // bench matrix is by itself an average workgroup's share; on a rank of an 8-GPU run it is eight shares).  A row
  // whose cost exceeds half a share is cut into up to kNeMaxSeg runs of whole steps of about a quarter share; the
  // workgroups that get the leading segments write their partial accumulators to an HBM scratch, the one with the
  // last segment adds them in segment order and solves (wrmf_ne.hip).  List entry >= 0: a row; -(s + 1): segment s
  // of the table {row, first non-zero, non-zeros, index within the row, segments of the row, scratch slot}.
  auto steps_of = [](int64_t len) { return (len + 15) / 16; };
  int64_t total = 0;
  for (int r = 0; r < n_prefix; r++) total += steps_of(len_of(r)) + fixed;
  const int64_t share = std::max<int64_t>(1, total / n_slots);
  struct Item { int64_t cost; int32_t entry; };
  std::vector<Item> items;
  items.reserve((size_t)n_prefix + 64);
  std::vector<int32_t> segs;
  int n_seg = 0;
  for (int r = 0; r < n_prefix; r++) {
    const int64_t len = len_of(r);
    const int64_t st = steps_of(len);
    int parts = 1;
    if (n_slots >= 8 && 2 * (st + fixed) > share)
      parts = (int)std::min<int64_t>(std::min<int64_t>(kNeMaxSeg, st / 64), (4 * st + share - 1) / share);   // (a segment: >= 64 steps)
    if (parts < 2 || n_seg + parts > kNeMaxSegTotal) {
      items.push_back({st + fixed, order[(size_t)r]});
      continue;
    }
    const int64_t per = (st + parts - 1) / parts;   // steps per segment
    const int slot = n_seg;
    int made = 0;
    for (int64_t s0 = 0; s0 < st; s0 += per, made++) {}
    int idx = 0;
    for (int64_t s0 = 0; s0 < st; s0 += per, idx++) {
      const int64_t n0 = s0 * 16, n1 = std::min(len, (s0 + per) * 16);
      segs.insert(segs.end(), {order[(size_t)r], (int32_t)n0, (int32_t)(n1 - n0), idx, made, slot});
      items.push_back({steps_of(n1 - n0) + fixed, -(int32_t)(n_seg + 1)});
      n_seg++;
    }
  }
  std::stable_sort(items.begin(), items.end(), [](const Item& x, const Item& y) { return x.cost > y.cost; });
  const size_t n_items = items.size();
  n_wg = (int)std::min<size_t>((size_t)n_wg, n_items);   // (no empty lists)
  std::vector<int> owner(n_items);
  std::vector<int32_t> cnt_wg((size_t)n_wg + 1, 0);
  std::priority_queue<std::pair<int64_t, int>, std::vector<std::pair<int64_t, int>>, std::greater<>> heap;
  for (int w = 0; w < n_wg; w++) heap.push({0, w});
  for (size_t e = 0; e < n_items; e++) {
    auto top = heap.top();
    heap.pop();
    owner[e] = top.second;
    cnt_wg[(size_t)top.second + 1]++;
    heap.push({top.first + items[e].cost, top.second});
  }
  for (int w = 0; w < n_wg; w++) cnt_wg[(size_t)w + 1] += cnt_wg[(size_t)w];
  std::vector<int32_t> lists(n_items), fill(cnt_wg.begin(), cnt_wg.end() - 1);
  for (size_t e = 0; e < n_items; e++) lists[(size_t)fill[(size_t)owner[e]]++] = items[e].entry;
  HIP_TRY(hipMalloc(&L.rows, lists.size() * sizeof(int32_t)));
  HIP_TRY(hipMemcpy(L.rows, lists.data(), lists.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&L.ptr, cnt_wg.size() * sizeof(int32_t)));
  HIP_TRY(hipMemcpy(L.ptr, cnt_wg.data(), cnt_wg.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  if (n_seg > 0) {
    HIP_TRY(hipMalloc(&L.segs, segs.size() * sizeof(int32_t)));
    HIP_TRY(hipMemcpy(L.segs, segs.data(), segs.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    std::vector<int32_t> srows, sptr{0};
    for (int sg = 0; sg < n_seg; sg++)
      if (segs[(size_t)sg * 6 + 3] == 0) {   // first segment of its row
        srows.push_back(-(int32_t)(sg + 1));
        sptr.push_back((int32_t)srows.size());
      }
    HIP_TRY(hipMalloc(&L.split_rows, srows.size() * sizeof(int32_t)));
    HIP_TRY(hipMemcpy(L.split_rows, srows.data(), srows.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&L.split_ptr, sptr.size() * sizeof(int32_t)));
    HIP_TRY(hipMemcpy(L.split_ptr, sptr.data(), sptr.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    L.nsplit = (int)srows.size();
  }
    L.nseg = n_seg;
    L.entries = (int)n_items;
    L.wg = n_wg;
  }
  return RSPARSE_HIP_OK;
}

// every row, longest first, with the bucket boundaries of the quad-layout CG kernels
int build_q_schedule(DevCSC& d, const int32_t* host_col_ptrs) {
  const int n = d.n_cols;
  for (int b = 0; b < 7; b++) d.q_off[b] = 0;
  for (int b = 0; b < 6; b++) d.q_nnz[b] = 0;
  d.q_order = nullptr;
  d.q_stream_off = nullptr;
  d.q_ne_rows = nullptr; d.q_ne_ptr = nullptr; d.q_ne_wg = 0;
  d.q_ne1_rows = nullptr; d.q_ne1_ptr = nullptr; d.q_ne1_wg = 0;
  d.q_ne_segs = nullptr; d.q_ne_nseg = 0; d.q_ne_entries = 0;
  d.q_ne_split_rows = nullptr; d.q_ne_split_ptr = nullptr; d.q_ne_nsplit = 0;
  d.q_nec_rows = nullptr; d.q_nec_ptr = nullptr; d.q_nec_wg = 0; d.q_nec_segs = nullptr; d.q_nec_nseg = 0; d.q_nec_entries = 0;
  d.q_nec_split_rows = nullptr; d.q_nec_split_ptr = nullptr; d.q_nec_nsplit = 0; d.q_nec_own = false; d.q_nec_min = kNeCholMinLen;
  d.q_n_chol_long = 0;
  d.q_lr_first = 0; d.q_n_lr = 0; d.q_gt32 = 0; d.q_gt48 = 0;
  d.q_pair_first = 0;
  d.q_cfg = cgq_default_cfg();
  if (n <= 0) return RSPARSE_HIP_OK;
  const int max_len = d.max_len;
  std::vector<int64_t> start((size_t)max_len + 2, 0);
  int cnt_b[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; i++) {
    const int len = host_col_ptrs[i + 1] - host_col_ptrs[i];
    start[(size_t)(max_len - len) + 1]++;
    const int b = cgq_bucket_of(len, d.q_cfg);
    if (len > kCholLongLen) d.q_n_chol_long++;
    if (len > 16) d.q_pair_first++;                // the order is longest first: the rows of <= 16 non-zeros are a suffix
    if (len > 32) d.q_gt32++;
    if (len > 48) d.q_gt48++;
    if (len > kCholLrMax) d.q_lr_first++;          // the order is longest first: the short rows are a suffix
    else if (len >= 1) d.q_n_lr++;
    cnt_b[b]++;
    d.q_nnz[b] += len;
  }
  for (size_t b = 1; b < start.size(); b++) start[b] += start[b - 1];
  std::vector<int32_t> order((size_t)n);
  for (int i = 0; i < n; i++) {
    const int len = host_col_ptrs[i + 1] - host_col_ptrs[i];
    order[(size_t)start[(size_t)(max_len - len)]++] = i;
  }
  for (int b = 0; b < 6; b++) d.q_off[b + 1] = d.q_off[b] + cnt_b[b];
  HIP_TRY(hipMalloc(&d.q_order, (size_t)n * sizeof(int32_t)));
  HIP_TRY(hipMemcpy(d.q_order, order.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
  // streamed bucket (bucket 0): slot of each row's first non-zero in the per-sweep scratch
  const int n_stream = d.q_off[1];
  if (n_stream > 0) {
    std::vector<int64_t> soff((size_t)n_stream + 1, 0);
    for (int r = 0; r < n_stream; r++) {
      const int row = order[(size_t)r];
      soff[(size_t)r + 1] = soff[(size_t)r] + (host_col_ptrs[row + 1] - host_col_ptrs[row]);
    }
    HIP_TRY(hipMalloc(&d.q_stream_off, soff.size() * sizeof(int64_t)));
    HIP_TRY(hipMemcpy(d.q_stream_off, soff.data(), soff.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    NeLists L;
    if (int rc2 = build_ne_lists(order, host_col_ptrs, n_stream, 12, L)) return rc2;
    d.q_ne_rows = L.rows; d.q_ne_ptr = L.ptr; d.q_ne_wg = L.wg; d.q_ne_segs = L.segs; d.q_ne_nseg = L.nseg;
    d.q_ne_entries = L.entries; d.q_ne_split_rows = L.split_rows; d.q_ne_split_ptr = L.split_ptr; d.q_ne_nsplit = L.nsplit;
    // the same rows and segments as one list per workgroup slot, for the ranks whose kernel is resident once per CU
    NeLists L1;
    if (int rc2 = build_ne_lists(order, host_col_ptrs, n_stream, 12, L1, false)) return rc2;
    d.q_ne1_rows = L1.rows; d.q_ne1_ptr = L1.ptr; d.q_ne1_wg = L1.wg;
    if (L1.segs) (void)hipFree(L1.segs);   // (identical to the tables above: the deal does not change the cut)
    if (L1.split_rows) (void)hipFree(L1.split_rows);
    if (L1.split_ptr) (void)hipFree(L1.split_ptr);
  }
  // a second set of lists for a longer prefix of the order.  Round 6: the rows beyond kCgMfMax non-zeros -- what is left to the
  // normal-equation kernel when wrmf_cg_mf.hip takes the rows of 513..kCgMfMax (rank 128, implicit conjugate gradient).
  // (Rounds 2-5: the rows beyond 64, for solver == CHOLESKY at rank 65..128 -- wrmf_chol_mf.hip has those now; dev builds
  // still set that threshold with RSPARSE_HIP_NE_CHOL_MIN = 64..512 for A/B runs of the old routing.)
  d.q_nec_min = kCgMfMax;
#ifdef RSP_AB
  if (const char* e = std::getenv("RSPARSE_HIP_NE_CHOL_MIN")) d.q_nec_min = std::min(kNeMinLen, std::max(kCholLrMax, std::atoi(e)));
#endif
  int n_nec = 0;
  while (n_nec < n && host_col_ptrs[order[(size_t)n_nec] + 1] - host_col_ptrs[order[(size_t)n_nec]] > d.q_nec_min) n_nec++;
  d.q_n_nec = n_nec;
  if (n_nec == n_stream) {
    d.q_nec_rows = d.q_ne_rows; d.q_nec_ptr = d.q_ne_ptr; d.q_nec_wg = d.q_ne_wg; d.q_nec_segs = d.q_ne_segs;
    d.q_nec_nseg = d.q_ne_nseg; d.q_nec_entries = d.q_ne_entries; d.q_nec_split_rows = d.q_ne_split_rows;
    d.q_nec_split_ptr = d.q_ne_split_ptr; d.q_nec_nsplit = d.q_ne_nsplit;
  } else {
    NeLists L;
    if (int rc2 = build_ne_lists(order, host_col_ptrs, n_nec, d.q_nec_min < kNeMinLen ? 72 : 12, L)) return rc2;
    d.q_nec_own = true;
    d.q_nec_rows = L.rows; d.q_nec_ptr = L.ptr; d.q_nec_wg = L.wg; d.q_nec_segs = L.segs; d.q_nec_nseg = L.nseg;
    d.q_nec_entries = L.entries; d.q_nec_split_rows = L.split_rows; d.q_nec_split_ptr = L.split_ptr; d.q_nec_nsplit = L.nsplit;
  }
  return RSPARSE_HIP_OK;
}

// the register-resident quad-layout CG kernels need rank % 4 == 0 and 16-byte aligned factor matrices; other ranks take
// the LDS-tile kernels of wrmf_kernels.hip
bool use_cgq(int rank, const void* X, const void* Y) {
  return rank % 4 == 0 && ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y)) & 15) == 0;
}

int check_common(int n_rows, int n_cols, const void* col_ptrs, const void* row_indices, const void* values,
                 const void* X, const void* Y, int rank) {
  if (n_rows < 0 || n_cols < 0) return fail(RSPARSE_HIP_ERR_INVALID, "negative matrix dimension");
  if (!col_ptrs) return fail(RSPARSE_HIP_ERR_INVALID, "col_ptrs is NULL");
  if (!X || !Y) return fail(RSPARSE_HIP_ERR_INVALID, "X or Y is NULL");
  if (rank <= 0) return fail(RSPARSE_HIP_ERR_INVALID, "rank must be positive");
  if (rank > RSPARSE_HIP_MAX_RANK)
    return fail(RSPARSE_HIP_ERR_UNSUPPORTED, "rank > 256 is not on the device path");
  (void)row_indices; (void)values;
  return RSPARSE_HIP_OK;
}

// rows whose failure counts were taken off the device on behalf of a LATER reader (rsparse_hip_take_numeric_failures adds them)
thread_local int64_t g_fail_carry[2] = {0, 0};   // unresolved, re-solved by the general solver

// wrmf_implicit.hpp:108-109: a global bias below sqrt(eps) of the element type T is treated as zero (float: 3.45e-4,
// double: 1.49e-8; the device layer holds floats, the stateless *_double entry point passes dbl = true)
bool has_global_bias(double global_bias, bool dbl = false) {
  return global_bias >= std::sqrt(dbl ? DBL_EPSILON : (double)FLT_EPSILON);
}

// Ranks that are not a multiple of 4 run on zero-padded copies: a padded coordinate sees a right-hand side of 0
// against lambda_use on the diagonal (implicit: the ridge of the padded Gramian, written as 1) and stays 0.  With explicit
// feedback, lambda = 0 (the reference's default) and the exact solver that diagonal is 0: every padded system would be
// singular where the true one is not.  Those fits keep the true rank on the LDS-tile kernels.
bool zero_padding_is_neutral(bool implicit, unsigned solver, double lambda) {
  return implicit || solver != RSPARSE_SOLVER_CHOLESKY || lambda != 0.0;
}

int check_variant(unsigned solver, int with_biases, double global_bias, bool implicit = true) {
  if (solver > RSPARSE_SOLVER_NNLS) return fail(RSPARSE_HIP_ERR_INVALID, "unknown solver code");
  if (with_biases && implicit && solver == RSPARSE_SOLVER_CONJUGATE_GRADIENT)
    // the reference drops a row of the warm start twice on this path (wrmf_implicit.hpp:189,197) and cannot run it
    return fail(RSPARSE_HIP_ERR_UNSUPPORTED, "with_user_item_bias + conjugate_gradient with implicit feedback is not on the device path");
  // (a global bias goes with every solver: Cholesky / NNLS wrmf_implicit.hpp:228-229,262-270, conjugate gradient
  // cg_solver_implicit_global_bias :35-57; with explicit feedback it is removed from the data by the R driver)
  (void)global_bias;
  return RSPARSE_HIP_OK;
}

struct BiasTerms {   // implicit feedback with user/item biases and / or a global bias, see AlsArgs
  const float* rhs_vals;
  const float* loss_tgt;
  const float* rhs_init;
  float tgt_const = 1.f;
  float gbias = 0.f;   // conjugate gradient with a global bias: the bias itself (rhs_init = global_bias_base)
};

// launch_ne_stats for the values of handle d: a handle whose values are frozen (rsparse_hip_csc_freeze_values) is scanned once
hipError_t take_value_stats(const DevCSC& d, const float* d_X, int64_t nx, hipStream_t s, const float* d_absmax) {
  if (d.vals_frozen && !d.vstats && hipMalloc(&d.vstats, 2 * sizeof(unsigned)) != hipSuccess) d.vstats = nullptr;
  // (the saved words are ordered behind their scan only on the stream that ran it: a call on another stream scans again and
  // takes the cache over -- ADVICE r05)
  if (d.vals_frozen && d.vstats && d.vstats_valid && d.vstats_stream == s)
    return launch_ne_stats(d_X, nx, d.vals, d.nnz, g_ws.ne_stats, s, d_absmax, d.vstats, nullptr);
  hipError_t e = launch_ne_stats(d_X, nx, d.vals, d.nnz, g_ws.ne_stats, s, d_absmax, nullptr,
                                 (d.vals_frozen && d.vstats) ? d.vstats : nullptr);
  if (e == hipSuccess && d.vals_frozen && d.vstats) { d.vstats_valid = true; d.vstats_stream = s; }
  return e;
}

// d_absmax (nullable, device float): max |X| supplied by the caller -- X is then not scanned for the fp16 operand scales
int run_half_iteration(const rsparse_hip_csc* conf, bool implicit, const float* d_X, float* d_Y,
                       const float* d_XtX, int rank, double lambda, unsigned solver, unsigned cg_steps,
                       int dynamic_lambda, double* d_loss_rows_out, hipStream_t s, const BiasTerms* bias = nullptr,
                       const float* d_absmax = nullptr) {
  if (!conf) return fail(RSPARSE_HIP_ERR_INVALID, "conf is NULL");
  if (!d_X || !d_Y) return fail(RSPARSE_HIP_ERR_INVALID, "X or Y is NULL");
  if (implicit && !d_XtX) return fail(RSPARSE_HIP_ERR_INVALID, "XtX is NULL");
  if (rank <= 0) return fail(RSPARSE_HIP_ERR_INVALID, "rank must be positive");
  if (rank > RSPARSE_HIP_MAX_RANK) return fail(RSPARSE_HIP_ERR_UNSUPPORTED, "rank > 256 is not on the device path");
  int rc = check_variant(solver, 0, 0.0);
  if (rc) return rc;
  if ((rc = g_ws.ensure_device())) return rc;
  const DevCSC& d = conf->d;
  // (end of round 6) the exact solver at the ranks 65..127: the rank-128 kernels (matrix-core assembly, a wave per row) on copies
  // padded to 128 beat the routes of a rank below the padded rank by 1.5-5 x (1M x 100k, ms per iteration: explicit rank 100
  // 102 -> 21, implicit 44 -> 29, explicit rank 64 with biases -- a system of order 65 -- 99 -> 28).  A padded coordinate is an exact
  // zero of an exact solve (identity / lambda on its diagonal, zero right-hand side); without per-non-zero bias operands only
  const bool pad_to_128 = solver == RSPARSE_SOLVER_CHOLESKY && !bias && rank > 64 && rank < 128 && zero_padding_is_neutral(implicit, solver, lambda);
  // ... and below rank 64 on copies padded to 64: the one-wave-per-row kernel with its matrix-core assembly exists at the padded rank
  // 64 only (1M x 100k, ms per iteration: rank 10 35 -> 20, rank 20 50 -> 20, rank 32 36 -> 20, rank 48 26 -> 20).  (First measured
  // with a failing cell of the reference's grid -- explicit feedback, lambda = 1000, the factors at 1e-28 --: that was the
  // explicit low-rank kernel's scaled solve overflowing, a bug of the native ranks 64 and 128 too, fixed in wrmf_chol_lr.hip.)
  const bool pad_to_64 = solver == RSPARSE_SOLVER_CHOLESKY && !bias && rank < 64 && zero_padding_is_neutral(implicit, solver, lambda);
  if (((rank % 4 != 0 && rank < 128) || pad_to_128 || pad_to_64) && zero_padding_is_neutral(implicit, solver, lambda)) {
    // A rank that is not a multiple of 4 (the reference's default is 10): the register-resident kernels take their vectors in
    // 16-byte pieces, and the LDS-tile fallback that took these ranks through round 3 is several times slower.  Coordinates of
    // zeros change nothing (wrmf_bias.hip, launch_pad_rows: the same argument as for the biased half-iterations), so the
    // half-iteration runs on copies padded to the next multiple of 4 and the solved rows are copied back.
    const int kp = pad_to_128 ? 128 : (pad_to_64 ? 64 : ((rank + 3) & ~3));
    const size_t nx = (size_t)d.n_rows * kp, ny = (size_t)d.n_cols * kp, ng = (size_t)kp * kp;
    if ((rc = g_ws.ensure_pad(nx + ny + ng + 256 + 16))) return rc;
    float* Xp = g_ws.pad_buf;
    float* Yp = Xp + nx;
    float* Gp = Yp + ny;
    float* rp = Gp + ng;
    hipError_t e;
    if ((e = launch_pad_rows(d_X, rank, 0, rank, kp, d.n_rows, Xp, s)) != hipSuccess) return hip_fail(e, "launch_pad_rows");
    if ((e = launch_pad_rows(d_Y, rank, 0, rank, kp, d.n_cols, Yp, s)) != hipSuccess) return hip_fail(e, "launch_pad_rows");
    if (implicit && (e = launch_pad_gramian(d_XtX, rank, kp, Gp, s)) != hipSuccess) return hip_fail(e, "launch_pad_gramian");
    BiasTerms bp;
    if (bias) {
      bp = *bias;
      if (bias->rhs_init) {   // (rank entries at the caller's: one row of kp, zeros beyond)
        if ((e = launch_pad_rows(bias->rhs_init, rank, 0, rank, kp, 1, rp, s)) != hipSuccess) return hip_fail(e, "launch_pad_rows");
        bp.rhs_init = rp;
      }
    }
    rc = run_half_iteration(conf, implicit, Xp, Yp, implicit ? Gp : nullptr, kp, lambda, solver, cg_steps, dynamic_lambda,
                            d_loss_rows_out, s, bias ? &bp : nullptr, d_absmax);
    if (rc) return rc;
    if (d.n_cols > 0)
      HIP_TRY(hipMemcpy2DAsync(d_Y, (size_t)rank * 4, Yp, (size_t)kp * 4, (size_t)rank * 4, (size_t)d.n_cols,
                               hipMemcpyDeviceToDevice, s));
    return RSPARSE_HIP_OK;
  }
  if (wide_supported(rank)) {
    // ranks 129..256: one kernel family for every solver and operand set (wrmf_wide.hip), no launch schedule
    double* outw = d_loss_rows_out ? d_loss_rows_out : g_ws.scalars;
    if (d.n_cols == 0) {
      HIP_TRY(hipMemsetAsync(outw, 0, sizeof(double), s));
      return RSPARSE_HIP_OK;
    }
    const int grid = wide_als_grid(d.n_cols);
    const int grid_w = wide_cg_wave_grid(d.n_cols) + wide_cg_team_grid(d.n_cols);   // (plain conjugate gradient: wrmf_wide_cg.hip's two launches)
    if ((rc = g_ws.ensure_partials((size_t)grid + (size_t)grid_w))) return rc;
    if ((rc = g_ws.ensure_wide(solver == RSPARSE_SOLVER_NNLS ? (size_t)grid * wide_m2_floats_per_wg(rank) : 0,
                               solver == RSPARSE_SOLVER_CHOLESKY ? (size_t)grid * rank * rank : 0)))
      return rc;
    AlsArgs a{};
    a.col_ptrs = d.col_ptrs; a.row_idx = d.row_idx; a.vals = d.vals;
    a.X = d_X; a.Y = d_Y; a.XtX = implicit ? d_XtX : nullptr;
    a.n_cols = d.n_cols; a.k = rank; a.cg_steps = (int)cg_steps;
    a.lambda = (float)lambda; a.lambda_loss = lambda; a.dynamic_lambda = dynamic_lambda ? 1 : 0;
    a.loss_partials = g_ws.partials; a.fail_counter = g_ws.fails;
    a.rhs_vals = bias ? bias->rhs_vals : nullptr;
    a.loss_tgt = bias ? bias->loss_tgt : nullptr;
    a.rhs_init = bias ? bias->rhs_init : nullptr;
    a.loss_tgt_const = bias ? bias->tgt_const : 1.f;
    a.gbias = (solver == RSPARSE_SOLVER_CONJUGATE_GRADIENT && implicit && bias) ? bias->gbias : 0.f;
    // plain conjugate gradient: the operator form, one wave per row (teams of 8 for the rows beyond kWideCgMaxLen non-zeros) -- no k x k
    // system; everything else: wrmf_wide.hip
    const bool wave_cg = wide_cg_wave_supported(a, solver) && (d.q_order || d.max_len <= kWideCgMaxLen);
    hipError_t we = wave_cg ? launch_wide_cg_wave(a, implicit, kWideCgMaxLen, d.max_len > kWideCgMaxLen ? d.q_order : nullptr, 0, s)
                            : launch_als_wide(a, implicit, solver, g_ws.wide_m2, g_ws.wide_lu, s);
    if (we != hipSuccess) return hip_fail(we, wave_cg ? "launch_wide_cg_wave" : "launch_als_wide");
    if ((we = launch_sum_partials(g_ws.partials, wave_cg ? (size_t)grid_w : (size_t)grid, outw, s, g_ws.partials + g_ws.partial_slots)) != hipSuccess)
      return hip_fail(we, "launch_sum_partials");
    return RSPARSE_HIP_OK;
  }
  const bool cg = solver == RSPARSE_SOLVER_CONJUGATE_GRADIENT;
  const bool cgq = cg && use_cgq(rank, d_X, d_Y);
  QSchedule qs;
  qs.order = d.q_order;
  qs.cfg = d.q_cfg;
  qs.pair_first = d.q_pair_first;
  for (int b = 0; b < 7; b++) qs.off[b] = d.q_off[b];
  // (two workgroups per CU only for implicit feedback at rank 97..128: the fp16 QUAD kernel of wrmf_ne.hip)
  const bool ne_fine = implicit && padded_rank(rank) == 128;
  qs.ne_rows = ne_fine ? d.q_ne_rows : d.q_ne1_rows; qs.ne_ptr = ne_fine ? d.q_ne_ptr : d.q_ne1_ptr;
  qs.ne_wg = ne_fine ? d.q_ne_wg : d.q_ne1_wg; qs.ne_entries = d.q_ne_entries;
  qs.ne_split_rows = d.q_ne_split_rows; qs.ne_split_ptr = d.q_ne_split_ptr; qs.ne_nsplit = d.q_ne_nsplit;
  // round 6, rank 128, implicit conjugate gradient without bias operands: the rows of 513..kCgMfMax non-zeros of the first bucket on
  // the wave-per-row kernel of wrmf_cg_mf.hip, the giant rows (a prefix of the order) on the normal-equation kernel's second lists
  bool cgm_on = cgq && implicit && !bias && cg_mf_supported(rank, implicit) && d.q_nec_min == kCgMfMax;
#ifdef RSP_AB
  if (const char* e = std::getenv("RSPARSE_HIP_CG_MF")) cgm_on = cgm_on && std::atoi(e) != 0;
#endif
  const bool cg_mf = cgm_on && d.q_order && d.q_off[1] > d.q_n_nec;
  if (cg_mf) {
    qs.ne_rows = d.q_nec_rows; qs.ne_ptr = d.q_nec_ptr; qs.ne_wg = d.q_nec_wg; qs.ne_entries = d.q_nec_entries;
    qs.ne_split_rows = d.q_nec_split_rows; qs.ne_split_ptr = d.q_nec_split_ptr; qs.ne_nsplit = d.q_nec_nsplit;
    qs.mf_rows = d.q_order + d.q_n_nec;
    qs.mf_n = d.q_off[1] - d.q_n_nec;
  }
  // solver == CHOLESKY: the rows beyond 512 non-zeros are assembled by the normal-equation kernel (matrix cores, one pass)
  // and solved exactly there (a.ne_chol); wrmf_chol.hip's kernel then skips them
  // (the lower threshold pays at rank 65..128 only: at rank <= 64 wrmf_chol.hip's kernel is cheaper than the fixed cost of
  // the tile solve up to 512 non-zeros -- config 5 with Cholesky 4.5 against 4.1 iterations/s)
  // (round 6, rank 65..128: the rows of 65..512 non-zeros go to the wave-per-row kernel of wrmf_chol_mf.hip instead -- the
  // normal-equation launch keeps the rows beyond 512, its CG lists; RSPARSE_HIP_CHOL_MF=0 in -DRSP_AB builds: the round-5 routing)
  bool mf_on = chol_mf_supported(rank);
#ifdef RSP_AB
  if (const char* e = std::getenv("RSPARSE_HIP_CHOL_MF")) mf_on = mf_on && std::atoi(e) != 0;
#endif
  const bool chol_mf = mf_on && !cg && solver == RSPARSE_SOLVER_CHOLESKY && !bias && d.q_order && ne_supported(rank) &&
                       use_cgq(rank, d_X, d_Y) && d.q_lr_first > d.q_off[1];
  int n_mf = chol_mf ? d.q_lr_first - d.q_off[1] : 0;
  int mf_first = d.q_off[1];
#ifdef RSP_AB   // timing experiments: every row beyond 64 non-zeros on the wave-per-row kernel (giant rows included: one wave each)
  if (chol_mf && std::getenv("RSPARSE_HIP_CHOL_MF_ALL")) {   // (= 2: without the rows beyond 4096 non-zeros, which nobody solves then)
    mf_first = std::atoi(std::getenv("RSPARSE_HIP_CHOL_MF_ALL")) == 2 ? d.q_n_chol_long : 0;
    n_mf = d.q_lr_first - mf_first;
  }
#endif
  const bool nec_lists = padded_rank(rank) > 64 && d.q_nec_wg > 0 && !chol_mf;
  const bool ne_chol = !cg && solver == RSPARSE_SOLVER_CHOLESKY && !bias && (nec_lists || d.q_ne_wg > 0 || chol_mf) &&
                       ne_supported(rank) && use_cgq(rank, d_X, d_Y);
  if (ne_chol && nec_lists) {   // its own lists: the rows beyond d.q_nec_min non-zeros
    qs.ne_rows = d.q_nec_rows; qs.ne_ptr = d.q_nec_ptr; qs.ne_wg = d.q_nec_wg; qs.ne_entries = d.q_nec_entries;
    qs.ne_split_rows = d.q_nec_split_rows; qs.ne_split_ptr = d.q_nec_split_ptr; qs.ne_nsplit = d.q_nec_nsplit;
  }
  const int ne_nseg = ((ne_chol && nec_lists) || cg_mf) ? d.q_nec_nseg : d.q_ne_nseg;
  const int32_t* ne_segs = ((ne_chol && nec_lists) || cg_mf) ? d.q_nec_segs : d.q_ne_segs;
  const size_t chol_base = chol2_loss_slots(d.n_cols);
  const size_t slots = cgq ? cgq_loss_slots(qs, rank, implicit) : (cg ? cg_loss_slots(d.n_cols, d.n_long)
                                   : (solver == RSPARSE_SOLVER_NNLS ? chol_loss_slots(d.n_cols)
                                      : chol_base + (ne_chol ? (size_t)(qs.ne_entries + qs.ne_nsplit) : 0) +
                                            (size_t)(chol_mf ? chol_mf_loss_slots(n_mf, implicit) : 0) + (size_t)kLuGrid));
  if ((rc = g_ws.ensure_partials(slots))) return rc;
  double* out = d_loss_rows_out ? d_loss_rows_out : g_ws.scalars;
  if (d.n_cols == 0) {
    HIP_TRY(hipMemsetAsync(out, 0, sizeof(double), s));
    return RSPARSE_HIP_OK;
  }
  AlsArgs a;
  a.col_ptrs = d.col_ptrs; a.row_idx = d.row_idx; a.vals = d.vals;
  a.X = d_X; a.Y = d_Y; a.XtX = implicit ? d_XtX : nullptr;
  a.long_rows = d.long_rows; a.n_long = d.n_long; a.n_cols = d.n_cols;
  a.chol_long_rows = d.q_order; a.n_chol_long = d.q_order ? d.q_n_chol_long : 0;
  a.chol_list = nullptr; a.chol_first = 0; a.chol_n_main = 0; a.chol_empty_first = 0;
  a.lr_rows = nullptr; a.n_lr = 0; a.lr_flags = nullptr; a.lr_M = nullptr; a.lr_n_gt32 = a.lr_n_gt16 = a.lr_n_gt48 = -1; a.lrx = 0;
  a.nnls_order = solver == RSPARSE_SOLVER_NNLS ? d.q_order : nullptr;
  a.nnls_lhs = nullptr;
  if (solver == RSPARSE_SOLVER_NNLS && padded_rank(rank) == 128) {   // (the wide family's buffer: no rank uses both)
    if ((rc = g_ws.ensure_wide(chol_loss_slots(d.n_cols) * (size_t)128 * 128, 0))) return rc;
    a.nnls_lhs = g_ws.wide_m2;
  }
  a.k = rank; a.cg_steps = (int)cg_steps;
  a.lambda = (float)lambda; a.lambda_loss = lambda; a.dynamic_lambda = dynamic_lambda ? 1 : 0;
  a.loss_partials = g_ws.partials; a.fail_counter = g_ws.fails; a.zero_row = g_ws.zero_row;
  a.fail_rows = g_ws.fails + 4; a.fail_cap = kFailCap;
  a.rhs_vals = bias ? bias->rhs_vals : nullptr;
  a.loss_tgt = bias ? bias->loss_tgt : nullptr;
  a.rhs_init = bias ? bias->rhs_init : nullptr;
  a.loss_tgt_const = bias ? bias->tgt_const : 1.f;
  const bool gb_cg = cg && implicit && bias && bias->gbias != 0.f;   // cg_solver_implicit_global_bias
  a.gbias = gb_cg ? bias->gbias : 0.f;
  a.ne_r0 = nullptr; a.ne_r0_slot = nullptr;
  a.tscr = nullptr; a.stream_off = d.q_stream_off; a.stream_nnz = d.q_nnz[0];
  a.ne_prof = nullptr;
  a.ne_stats = nullptr; a.wave_stats = nullptr; a.mf_XtX = nullptr;
  a.ne_segs = nullptr; a.ne_seg_scratch = nullptr; a.ne_seg_flags = nullptr;
  a.ne_chol = ne_chol ? 1 : 0;
  a.ne_chol_min = nec_lists ? d.q_nec_min : kNeMinLen;
  if (d.q_order && solver == RSPARSE_SOLVER_CHOLESKY) {
    // the k x k kernel's own rows as ranges of the length-sorted order: behind the prefix that the normal-equation launch
    // (or the LONG launch) takes, down to the short rows of the low-rank kernel, and the empty rows at the end
    const int first = chol_mf ? d.q_lr_first : (ne_chol ? (nec_lists ? d.q_n_nec : d.q_off[1]) : d.q_n_chol_long);
    a.chol_list = d.q_order;
    a.chol_first = first;
    a.chol_n_main = std::max(0, d.q_lr_first - first);
    a.chol_empty_first = d.q_lr_first + d.q_n_lr;
  }
  if ((cgq || ne_chol) && ne_nseg > 0 && ne_supported(rank)) {
    if ((rc = g_ws.ensure_ne_seg((size_t)ne_nseg))) return rc;
    a.ne_segs = ne_segs; a.ne_seg_scratch = g_ws.ne_seg_scratch; a.ne_seg_flags = g_ws.ne_seg_flags;
  }
  if (chol_mf) {   // the operand scales of its matrix-core assembly (max |X|, max c)
    hipError_t se = take_value_stats(d, d_X, (int64_t)d.n_rows * rank, s, d_absmax);
    if (se != hipSuccess) return hip_fail(se, "launch_ne_stats");
    a.wave_stats = g_ws.ne_stats;
    if (implicit) {
      a.ne_stats = g_ws.ne_stats;
      a.mf_XtX = d_XtX;
      if (rank != 128) {
        if ((rc = g_ws.ensure_mf())) return rc;
        hipError_t pe = launch_pad_gramian(d_XtX, rank, 128, g_ws.mf_G, s);
        if (pe != hipSuccess) return hip_fail(pe, "launch_pad_gramian");
        a.mf_XtX = g_ws.mf_G;
      }
    }
  }
  if (!a.ne_stats && (cgq || ne_chol) && implicit && (qs.ne_wg > 0 || qs.mf_n > 0) && ne_supported(rank) && (!bias || gb_cg)) {
    // operand scales of the fp16 normal-equation kernel (and whether it may run at all), decided on the device
    hipError_t se = take_value_stats(d, d_X, (int64_t)d.n_rows * rank, s, d_absmax);
    if (se != hipSuccess) return hip_fail(se, "launch_ne_stats");
    a.ne_stats = g_ws.ne_stats;
    if (gb_cg && cgq) {   // the long rows' share of the first residual: base - g X_nnz (c - 1), one more pass over them
      const int n_ne = d.q_off[1];
      if ((rc = g_ws.ensure_gb((size_t)n_ne * rank, (size_t)d.n_cols))) return rc;
      hipError_t ge = launch_gb_row_terms(a, d.q_order, n_ne, g_ws.gb_r0, g_ws.gb_slot, s);
      if (ge != hipSuccess) return hip_fail(ge, "launch_gb_row_terms");
      a.ne_r0 = g_ws.gb_r0; a.ne_r0_slot = g_ws.gb_slot;
    }
  }
#ifdef RSP_NE_PROF
  static unsigned long long* prof_buf = nullptr;
  if (!prof_buf) { HIP_TRY(hipMalloc(&prof_buf, (size_t)65536 * 4 * 20 * 8)); }
  HIP_TRY(hipMemsetAsync(prof_buf, 0, (size_t)65536 * 4 * 20 * 8, s));
  a.ne_prof = prof_buf;
#endif
  // ranks the normal-equation kernel does not take (<= 32): the streamed CG bucket keeps its per-sweep dot products in
  // an HBM scratch instead of re-gathering for the loss
  if (cgq && !ne_supported(rank) && d.q_stream_off && d.q_nnz[0] > 0 && cg_steps >= 1 && cg_steps <= 4) {
    if ((rc = g_ws.ensure_tscr((size_t)(cg_steps + 1) * (size_t)d.q_nnz[0]))) return rc;
    a.tscr = g_ws.tscr;
  }
  if (!cg && solver == RSPARSE_SOLVER_CHOLESKY && d.q_order && d.q_n_lr > 0 && chol_lr_supported(a, implicit)) {
    // low-rank form for the short rows; "some confidence < 1" comes from the values scan of launch_ne_stats (word 2)
    if ((rc = g_ws.ensure_lr())) return rc;
    if (!a.ne_stats) {   // (with the long rows on the normal-equation kernel the full statistics were just taken)
      // max |X| too: the low-rank kernel scales its fp16 operand terms by it
      hipError_t se = take_value_stats(d, d_X, (int64_t)d.n_rows * rank, s, d_absmax);
      if (se != hipSuccess) return hip_fail(se, "launch_ne_stats");
    }
    a.lr_rows = d.q_order + d.q_lr_first; a.n_lr = d.q_n_lr; a.lr_flags = g_ws.ne_stats + 2; a.lr_M = g_ws.lr_M;
    a.lr_n_gt32 = d.q_gt32 - d.q_lr_first; a.lr_n_gt16 = d.q_pair_first - d.q_lr_first;
    a.lr_n_gt48 = d.q_gt48 - d.q_lr_first;
  }
  const bool chol = !cg && solver == RSPARSE_SOLVER_CHOLESKY;
  if (chol && d.q_order && d.q_n_lr > 0 && chol_lrx_supported(a, implicit)) {
    // explicit feedback, ranks 64 / 128: the rows of 1..64 ratings in push-through form, one wave per pass
    if (!a.ne_stats && !a.wave_stats) {
      hipError_t se = take_value_stats(d, d_X, (int64_t)d.n_rows * rank, s, d_absmax);
      if (se != hipSuccess) return hip_fail(se, "launch_ne_stats");
    }
    a.wave_stats = g_ws.ne_stats;
    a.lr_rows = d.q_order + d.q_lr_first; a.n_lr = d.q_n_lr; a.lrx = 1;
    a.lr_n_gt32 = d.q_gt32 - d.q_lr_first; a.lr_n_gt16 = d.q_pair_first - d.q_lr_first;
    a.lr_n_gt48 = d.q_gt48 - d.q_lr_first;
  }
  if (chol && chol_wave_supported(rank) && padded_rank(rank) == 64) {
    // rank 33..64, one wave per row: the assembly runs on the matrix cores from fp16 operand terms scaled by max |X| (and max c)
    if (!a.ne_stats && !a.wave_stats) {
      hipError_t se = take_value_stats(d, d_X, (int64_t)d.n_rows * rank, s, d_absmax);
      if (se != hipSuccess) return hip_fail(se, "launch_ne_stats");
    }
    a.wave_stats = g_ws.ne_stats;
  }
  if (chol) {
    hipError_t fe = launch_fail_roll(g_ws.fails, s);
    if (fe != hipSuccess) return hip_fail(fe, "launch_fail_roll");
  }
  hipEvent_t* ev = g_prof.begin();
  if (chol && ev) HIP_TRY(hipEventRecord(ev[0], s));   // Cholesky: [0] normal-equation launch, [1] low-rank, [2] k x k, [3] wave-per-row (rank 65..128), [4] loss
  if (ne_chol && qs.ne_wg > 0 && mf_first != 0) {
    hipError_t ne = launch_als_ne(a, qs, implicit, g_ws.partials + chol_base, s, ev);
    if (ne != hipSuccess) return hip_fail(ne, "launch_als_ne");
  }
  hipError_t e = cgq ? launch_als_cgq(a, qs, implicit, s, ev)
                     : (cg ? launch_als_cg(a, implicit, s, ev)
                           : (solver == RSPARSE_SOLVER_NNLS
                                  ? launch_als_nnls(a, implicit, s, ev)
                                  : launch_als_chol2(a, implicit, s, ev ? ev + 1 : nullptr)));
  if (e != hipSuccess) return hip_fail(e, cgq ? "launch_als_cgq" : (cg ? "launch_als_cg" : "launch_als_chol"));
  if (chol_mf) {   // (launch_als_chol2 recorded ev[3] behind its kernels)
    const size_t mf_base = chol_base + (size_t)(qs.ne_entries + qs.ne_nsplit);
    hipError_t me = launch_als_chol_mf(a, implicit, d.q_order + mf_first, n_mf, (int)mf_base, s, ev ? ev + 3 : nullptr);
    if (me != hipSuccess) return hip_fail(me, "launch_als_chol_mf");
  }
  if (chol && ev) HIP_TRY(hipEventRecord(ev[4], s));
#if defined(RSP_NE_PROF) && defined(RSP_MF_PROF)
  if (chol_mf && std::getenv("RSPARSE_MF_PROF")) {   // phase ticks of wrmf_chol_mf.hip, summed over its waves
    HIP_TRY(hipStreamSynchronize(s));
    unsigned long long hp[8];
    HIP_TRY(hipMemcpy(hp, prof_buf + 8, sizeof(hp), hipMemcpyDeviceToHost));
    const char* nm[8] = {"zero", "assembly", "finalize", "forward", "backward", "fail/store", "loss", "row-head"};
    std::fprintf(stderr, "[mf_prof] n_cols %d rows %d: G ticks summed over the waves:", d.n_cols, n_mf);
    for (int j = 0; j < 8; j++) std::fprintf(stderr, " %s %.3f", nm[j], (double)hp[j] / 1e9);
    std::fprintf(stderr, "\n");
  }
#endif
  if (chol) {
    // rows whose factorisation met a non-positive pivot: the general solver, as arma::solve falls back to (wrmf_lu.hip).
    // The counters are per call: a previous call's count was taken by rsparse_hip_take_numeric_failures or is added to.
    if ((e = launch_als_lu_fallback(a, implicit, slots - (size_t)kLuGrid, s)) != hipSuccess)
      return hip_fail(e, "launch_als_lu_fallback");
  }
  e = launch_sum_partials(g_ws.partials, slots, out, s, g_ws.partials + g_ws.partial_slots);
  if (e != hipSuccess) return hip_fail(e, "launch_sum_partials");
#if defined(RSP_NE_PROF) && defined(RSP_NNLS_PROF)
  if (solver == RSPARSE_SOLVER_NNLS && std::getenv("RSPARSE_NNLS_PROF")) {   // phase ticks of als_nnls_wave_kernel, summed over its waves
    HIP_TRY(hipStreamSynchronize(s));
    unsigned long long hp[8];
    HIP_TRY(hipMemcpy(hp, prof_buf + 32, sizeof(hp), hipMemcpyDeviceToHost));
    std::fprintf(stderr, "[nnls_prof] n_cols %d: rows %llu, sweeps per row %.1f, coordinates visited per sweep %.1f; G ticks summed over the waves: assembly %.3f square %.3f sweeps %.3f loss %.3f row-head %.3f\n",
                 d.n_cols, hp[5], hp[5] ? (double)hp[6] / (double)hp[5] : 0.0, hp[6] ? (double)hp[7] / (double)hp[6] : 0.0,
                 (double)hp[0] / 1e9, (double)hp[1] / 1e9, (double)hp[2] / 1e9, (double)hp[3] / 1e9, (double)hp[4] / 1e9);
  }
#endif
#if defined(RSP_NE_PROF) && defined(RSP_MF_PROF)
  if (cgq && std::getenv("RSPARSE_MF_PROF")) {   // phase ticks of wrmf_cg_mf.hip, summed over its waves
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long hp[12];
    HIP_TRY(hipMemcpy(hp, prof_buf + 16, sizeof(hp), hipMemcpyDeviceToHost));
    const char* nm[8] = {"row-head", "prologue", "wait", "request", "step", "flush+unscale", "cg", "loss+tail"};
    std::fprintf(stderr, "[cgmf_prof] n_cols %d: rows %llu steps %llu, shader clock %.3f GHz (s_memtime / s_memrealtime), wave-seconds %.3f, G ticks summed over the waves (total %.3f):",
                 d.n_cols, hp[9], hp[8], hp[11] ? (double)hp[10] / (double)hp[11] * 0.1 : 0.0, (double)hp[11] / 1e8, (double)hp[10] / 1e9);
    for (int j = 0; j < 8; j++) std::fprintf(stderr, " %s %.3f", nm[j], (double)hp[j] / 1e9);
    std::fprintf(stderr, "\n");
  }
#endif
#ifdef RSP_NE_PROF
  if ((cgq || ne_chol) && qs.ne_wg > 0 && std::getenv("RSPARSE_NE_PROF")) {   // (solver == CHOLESKY: mv_* = diagonal tile, panel, trailing, backward)
    HIP_TRY(hipStreamSynchronize(s));
    std::vector<unsigned long long> hp((size_t)qs.ne_wg * 4 * 20);
    HIP_TRY(hipMemcpy(hp.data(), prof_buf, hp.size() * 8, hipMemcpyDeviceToHost));
    const char* nm[20] = {"wait_vm", "barrier", "issue", "consume", "tail", "total", "rows", "pre", "chain", "cg", "quad", "fin", "ch_sync", "ch_add", "ch_bar", "mv_pub", "mv_units", "mv_bar", "mv_comb", "-"};
    for (int w = 0; w < 4; w++) {
      std::fprintf(stderr, "[ne_prof] wave %d (n_cols %d):", w, d.n_cols);
      for (int j = 0; j < 19; j++) {
        double sum = 0, mx = 0;
        for (int b = 0; b < qs.ne_wg; b++) { const double v = (double)hp[((size_t)b * 4 + w) * 20 + j]; sum += v; mx = std::max(mx, v); }
        std::fprintf(stderr, " %s %.2f/%.2f", nm[j], sum / qs.ne_wg / 1e6, mx / 1e6);
      }
      std::fprintf(stderr, " Mcycles (mean/max over %d workgroups)\n", qs.ne_wg);
    }
#ifdef RSP_CGQ_PROF
    {
      unsigned long long cq[6];
      HIP_TRY(hipMemcpy(cq, prof_buf + (size_t)65536 * 80 - 16, sizeof(cq), hipMemcpyDeviceToHost));
      std::fprintf(stderr, "[cgq_prof] 8-wave launch (n_cols %d), G ticks summed over its waves: gather %.2f setup %.2f sweeps %.2f issue %.2f cg %.2f tail %.2f\n",
                   d.n_cols, cq[0] / 1e9, cq[1] / 1e9, cq[2] / 1e9, cq[3] / 1e9, cq[4] / 1e9, cq[5] / 1e9);
    }
#endif
    if (const char* dump = std::getenv("RSPARSE_NE_PROF_DUMP")) {   // per workgroup: total, rows, consume, wait_vm of wave 0
      if (FILE* f = std::fopen(dump, "a")) {
        std::fprintf(f, "# launch n_cols %d wgs %d\n", d.n_cols, qs.ne_wg);
        for (int b = 0; b < qs.ne_wg; b++)
          std::fprintf(f, "%d %llu %llu %llu %llu\n", b, hp[((size_t)b * 4) * 20 + 5], hp[((size_t)b * 4) * 20 + 6],
                       hp[((size_t)b * 4) * 20 + 3], hp[((size_t)b * 4) * 20 + 0]);
        std::fclose(f);
      }
    }
  }
#endif
  if (ev) {
    const int last = cgq ? 7 : (chol ? 5 : 3);
    HIP_TRY(hipEventRecord(ev[last], s));
    g_prof.have = true;
    g_prof.nseg = last;
  }
  return RSPARSE_HIP_OK;
}

template <class T>
std::vector<float> to_f32(const T* src, size_t n) {
  std::vector<float> v(n);
  for (size_t i = 0; i < n; i++) v[i] = (float)src[i];
  return v;
}

// als_explicit<T> with_biases (inst/include/wrmf_explicit.hpp:41-64,86-91,113-127) by re-packing: see wrmf_bias.hip.
int run_half_iteration_explicit_biased(const rsparse_hip_csc* conf, const float* d_X, float* d_Y, int rank, double lambda,
                                       unsigned solver, unsigned cg_steps, int dynamic_lambda, int is_x_bias_last_row,
                                       double* d_loss_rows_out, hipStream_t s) {
  if (!conf) return fail(RSPARSE_HIP_ERR_INVALID, "conf is NULL");
  if (!d_X || !d_Y) return fail(RSPARSE_HIP_ERR_INVALID, "X or Y is NULL");
  if (rank < 2) return fail(RSPARSE_HIP_ERR_INVALID, "with_biases needs rank >= 2 (a row of ones and a bias row)");
  if (rank > RSPARSE_HIP_MAX_RANK) return fail(RSPARSE_HIP_ERR_UNSUPPORTED, "rank > 256 is not on the device path");
  int rc = g_ws.ensure_device();
  if (rc) return rc;
  const DevCSC& d = conf->d;
  const int k1 = rank - 1;
  const int xoff = is_x_bias_last_row ? 0 : 1;       // first kept row of X_nnz            (:88)
  const int xb = is_x_bias_last_row ? rank - 1 : 0;  // row of X holding the x biases       (:59-64)
  const int ioff = is_x_bias_last_row ? 1 : 0;       // first kept entry of the warm start  (:90)
  const int ooff = is_x_bias_last_row ? 0 : 1;       // head / tail of Y.col(i)             (:115-127)
  // the copies are padded to a multiple of 4 (wrmf_bias.hip: why) -- unless the padding would make the systems singular
  const int k1p = zero_padding_is_neutral(false, solver, lambda) ? std::min((k1 + 3) & ~3, RSPARSE_HIP_MAX_RANK) : k1;
  const size_t nx = (size_t)d.n_rows * k1p, ny = (size_t)d.n_cols * k1p, nv = (size_t)std::max<int64_t>(d.nnz, 1);
  if ((rc = g_ws.ensure_bias(nx + ny + nv + 16))) return rc;
  float* Xp = g_ws.bias_buf;
  float* Yp = Xp + nx;
  float* vp = Yp + ny;
  hipError_t e;
  if ((e = launch_pad_rows(d_X, rank, xoff, k1, k1p, d.n_rows, Xp, s)) != hipSuccess) return hip_fail(e, "launch_pad_rows");
  if ((e = launch_pad_rows(d_Y, rank, ioff, k1, k1p, d.n_cols, Yp, s)) != hipSuccess) return hip_fail(e, "launch_pad_rows");
  e = launch_bias_shift_values(d.vals, d.row_idx, d_X, rank, xb, d.nnz, vp, s);
  if (e != hipSuccess) return hip_fail(e, "launch_bias_shift_values");
  rsparse_hip_csc shifted = *conf;   // same sparsity and schedule, shifted ratings (a view: never destroyed)
  shifted.d.vals_frozen = false; shifted.d.vstats_valid = false; shifted.d.vstats = nullptr;   // (its values are this call's)
  shifted.d.vals = vp;
  shifted.d.owns_matrix = false;
  rc = run_half_iteration(&shifted, false, Xp, Yp, nullptr, k1p, lambda, solver, cg_steps, dynamic_lambda,
                          d_loss_rows_out, s);
  if (rc) return rc;
  if (d.n_cols > 0)
    HIP_TRY(hipMemcpy2DAsync(d_Y + ooff, (size_t)rank * 4, Yp, (size_t)k1p * 4, (size_t)k1 * 4, (size_t)d.n_cols,
                             hipMemcpyDeviceToDevice, s));
  return RSPARSE_HIP_OK;
}

// als_implicit<T> with_biases, Cholesky / NNLS (inst/include/wrmf_implicit.hpp:114-154,186-252,256-270) by re-packing
// plus three extra operands of the solve kernels (AlsArgs::rhs_vals / loss_tgt / rhs_init).  d_XtX is the
// (rank-1) x (rank-1) Gramian of X without its x_bias row, ridge included (R/model_WRMF.R:474-486).
int run_half_iteration_implicit_biased(const rsparse_hip_csc* conf, const float* d_X, float* d_Y, const float* d_XtX,
                                       int rank, double lambda, unsigned solver, int is_x_bias_last_row,
                                       double* d_loss_rows_out, hipStream_t s, double global_bias = 0.0,
                                       bool dbl_threshold = false) {
  if (!conf) return fail(RSPARSE_HIP_ERR_INVALID, "conf is NULL");
  if (!d_X || !d_Y || !d_XtX) return fail(RSPARSE_HIP_ERR_INVALID, "X, Y or XtX is NULL");
  if (rank < 2) return fail(RSPARSE_HIP_ERR_INVALID, "with_biases needs rank >= 2 (a row of ones and a bias row)");
  if (rank > RSPARSE_HIP_MAX_RANK) return fail(RSPARSE_HIP_ERR_UNSUPPORTED, "rank > 256 is not on the device path");
  int rc = g_ws.ensure_device();
  if (rc) return rc;
  const DevCSC& d = conf->d;
  const int k1 = rank - 1;
  const int xoff = is_x_bias_last_row ? 0 : 1, xb = is_x_bias_last_row ? rank - 1 : 0;
  const int ioff = is_x_bias_last_row ? 1 : 0, ooff = is_x_bias_last_row ? 0 : 1;
  const int k1p = std::min((k1 + 3) & ~3, RSPARSE_HIP_MAX_RANK);   // the copies are padded to a multiple of 4 (wrmf_bias.hip: why)
  const size_t nx = (size_t)d.n_rows * k1p, ny = (size_t)d.n_cols * k1p, nv = (size_t)std::max<int64_t>(d.nnz, 1);
  const size_t nscr = bias_rhs_init_scratch_floats(), ng = (size_t)k1p * k1p;
  if ((rc = g_ws.ensure_bias(nx + ny + 2 * nv + nscr + ng + 16))) return rc;
  float* Xp = g_ws.bias_buf;
  float* Yp = Xp + nx;
  float* rcoef = Yp + ny;
  float* tgt = rcoef + nv;
  float* scratch = tgt + nv;
  float* rinit = scratch + (nscr - 256);   // (256 entries, zeros beyond k1)
  float* Gp = scratch + nscr;
  hipError_t e;
  if ((e = launch_pad_rows(d_X, rank, xoff, k1, k1p, d.n_rows, Xp, s)) != hipSuccess) return hip_fail(e, "launch_pad_rows");
  if ((e = launch_pad_rows(d_Y, rank, ioff, k1, k1p, d.n_cols, Yp, s)) != hipSuccess) return hip_fail(e, "launch_pad_rows");
  const float* G_use = d_XtX;
  if (k1p != k1) {
    if ((e = launch_pad_gramian(d_XtX, k1, k1p, Gp, s)) != hipSuccess) return hip_fail(e, "launch_pad_gramian");
    G_use = Gp;
  }
  const float gb = has_global_bias(global_bias, dbl_threshold) ? (float)global_bias : 0.f;
  e = launch_bias_implicit_terms(d.vals, d.row_idx, d_X, rank, xb, d.nnz, gb, rcoef, tgt, s);
  if (e != hipSuccess) return hip_fail(e, "launch_bias_implicit_terms");
  if ((e = launch_bias_rhs_init(d_X, rank, xoff, k1, xb, gb, d.n_rows, scratch, rinit, s)) != hipSuccess)
    return hip_fail(e, "launch_bias_rhs_init");
  BiasTerms bt{rcoef, tgt, rinit};
  rc = run_half_iteration(conf, true, Xp, Yp, G_use, k1p, lambda, solver, 0, 0, d_loss_rows_out, s, &bt);
  if (rc) return rc;
  if (d.n_cols > 0)
    HIP_TRY(hipMemcpy2DAsync(d_Y + ooff, (size_t)rank * 4, Yp, (size_t)k1p * 4, (size_t)k1 * 4, (size_t)d.n_cols,
                             hipMemcpyDeviceToDevice, s));
  return RSPARSE_HIP_OK;
}

// als_implicit<T> with a global bias and no user/item biases, Cholesky / NNLS (inst/include/wrmf_implicit.hpp:108-112,
// 155-157, 228-229, 262-264): rhs = X_nnz c + global_bias_base with global_bias_base = -global_bias * rowSums(X)
// (computed here when d_base_in is NULL -- initialize_bias_base -- and returned through d_base_out if given), every row
// is solved (empty ones too, :178), and the loss compares x_j.y with 1 - global_bias.
int run_half_iteration_implicit_global(const rsparse_hip_csc* conf, const float* d_X, float* d_Y, const float* d_XtX,
                                       int rank, double lambda, unsigned solver, unsigned cg_steps, double global_bias,
                                       const float* d_base_in, float* d_base_out, double* d_loss_rows_out,
                                       hipStream_t s, const float* d_absmax = nullptr) {
  if (!conf) return fail(RSPARSE_HIP_ERR_INVALID, "conf is NULL");
  if (!d_X || !d_Y || !d_XtX) return fail(RSPARSE_HIP_ERR_INVALID, "X, Y or XtX is NULL");
  if (rank <= 0) return fail(RSPARSE_HIP_ERR_INVALID, "rank must be positive");
  if (rank > RSPARSE_HIP_MAX_RANK) return fail(RSPARSE_HIP_ERR_UNSUPPORTED, "rank > 256 is not on the device path");
  int rc = g_ws.ensure_device();
  if (rc) return rc;
  const DevCSC& d = conf->d;
  const size_t nscr = bias_rhs_init_scratch_floats();
  if ((rc = g_ws.ensure_bias(nscr + 16))) return rc;
  float* scratch = g_ws.bias_buf;
  float* rinit = scratch + (nscr - 256);
  if (d_base_in) {
    HIP_TRY(hipMemcpyAsync(rinit, d_base_in, (size_t)rank * 4, hipMemcpyDeviceToDevice, s));
  } else {
    hipError_t e = launch_bias_rhs_init(d_X, rank, 0, rank, -1, (float)global_bias, d.n_rows, scratch, rinit, s);
    if (e != hipSuccess) return hip_fail(e, "launch_bias_rhs_init");
  }
  if (d_base_out) HIP_TRY(hipMemcpyAsync(d_base_out, rinit, (size_t)rank * 4, hipMemcpyDeviceToDevice, s));
  // conjugate gradient: cg_solver_implicit_global_bias (wrmf_implicit.hpp:35-57,203) from the warm start in d_Y
  const bool cg = solver == RSPARSE_SOLVER_CONJUGATE_GRADIENT;
  BiasTerms bt{nullptr, nullptr, rinit, (float)(1.0 - global_bias), cg ? (float)global_bias : 0.f};
  return run_half_iteration(conf, true, d_X, d_Y, d_XtX, rank, lambda, solver, cg ? cg_steps : 0, 0, d_loss_rows_out, s,
                            &bt, d_absmax);
}

// Shared body of the four stateless drop-ins.  TX = float or double (host element type).
template <class TX>
int stateless(bool implicit, int n_rows, int n_cols, const int32_t* col_ptrs, const int32_t* row_indices,
              const double* values, const TX* X, TX* Y, const TX* XtX, const TX* cnt_X, int rank, double lambda,
              unsigned solver, unsigned cg_steps, int dynamic_lambda, double* loss_out, int with_biases = 0,
              int is_x_bias_last_row = 0, double global_bias = 0.0, TX* global_bias_base = nullptr,
              int global_bias_base_len = 0, int initialize_bias_base = 1) {
  rsparse_hip_csc* conf = nullptr;
  int rc = rsparse_hip_csc_create_host(n_rows, n_cols, col_ptrs, row_indices, values, &conf);
  if (rc) return rc;
  struct Guard { rsparse_hip_csc* c; ~Guard() { rsparse_hip_csc_destroy(c); } } guard{conf};
  const size_t nx = (size_t)rank * n_rows, ny = (size_t)rank * n_cols;
  const size_t ng = implicit && with_biases ? (size_t)(rank - 1) * (rank - 1) : (size_t)rank * rank;
  DevBuf dX, dY, dG, dW;
  HIP_TRY(dX.alloc(nx * 4));
  HIP_TRY(dY.alloc(ny * 4));
  std::vector<float> tmp;
  auto upload = [&](DevBuf& b, const TX* src, size_t n) -> hipError_t {
    if (!n) return hipSuccess;
    if (sizeof(TX) == sizeof(float)) return hipMemcpy(b.p, src, n * 4, hipMemcpyHostToDevice);
    tmp = to_f32(src, n);
    return hipMemcpy(b.p, tmp.data(), n * 4, hipMemcpyHostToDevice);
  };
  HIP_TRY(upload(dX, X, nx));
  HIP_TRY(upload(dY, Y, ny));
  if (implicit) {
    if (!XtX) return fail(RSPARSE_HIP_ERR_INVALID, "XtX is NULL");
    HIP_TRY(dG.alloc(ng * 4));
    HIP_TRY(upload(dG, XtX, ng));
  }
  const bool weighted = !implicit && dynamic_lambda;
  if (weighted && lambda > 0) {
    if (!cnt_X) return fail(RSPARSE_HIP_ERR_INVALID, "cnt_X is NULL with dynamic_lambda");
    HIP_TRY(dW.alloc((size_t)n_rows * 4));
    HIP_TRY(upload(dW, cnt_X, (size_t)n_rows));
  }
  if ((rc = g_ws.ensure_device())) return rc;
  // counters left behind by earlier device-resident calls are not this call's: set aside here, handed back when this call
  // ends (a stateless call between a resident fit's half-iterations and its check must not swallow the fit's failures)
  StaleFailures stale_guard;
  const bool gbias = implicit && has_global_bias(global_bias, sizeof(TX) == sizeof(double));
  DevBuf dBase;
  if (gbias && !with_biases) {
    // global_bias_base = -global_bias * rowSums(X), `rank` entries (wrmf_implicit.hpp:111-112).  The caller's buffer holds
    // global_bias_base_len entries -- the R driver allocates rank - 1 (R/model_WRMF.R:292) although the reference's C++ reads
    // and assigns `rank`; here never more than the stated length is touched: it is READ (initialize_bias_base == 0) only
    // when it holds the whole vector, otherwise the vector is recomputed from X (its definition); it is WRITTEN up to
    // min(len, rank) entries
    HIP_TRY(dBase.alloc((size_t)rank * 4));
    const int blen = global_bias_base ? std::max(global_bias_base_len, 0) : 0;
    const bool given = !initialize_bias_base && blen >= rank;
    if (given) HIP_TRY(upload(dBase, global_bias_base, (size_t)rank));
    rc = run_half_iteration_implicit_global(conf, dX.as<float>(), dY.as<float>(), dG.as<float>(), rank, lambda, solver,
                                            cg_steps, global_bias, given ? dBase.as<float>() : nullptr,
                                            given ? nullptr : dBase.as<float>(), g_ws.scalars, nullptr);
    if (!rc && !given && initialize_bias_base && blen > 0) {
      std::vector<float> hb((size_t)rank);
      HIP_TRY(hipMemcpy(hb.data(), dBase.p, (size_t)rank * 4, hipMemcpyDeviceToHost));
      for (int t = 0; t < std::min(blen, rank); t++) global_bias_base[t] = (TX)hb[(size_t)t];
    }
  } else if (with_biases && implicit)
    rc = run_half_iteration_implicit_biased(conf, dX.as<float>(), dY.as<float>(), dG.as<float>(), rank, lambda, solver,
                                            is_x_bias_last_row, g_ws.scalars, nullptr, global_bias,
                                            sizeof(TX) == sizeof(double));
  else if (with_biases)
    rc = run_half_iteration_explicit_biased(conf, dX.as<float>(), dY.as<float>(), rank, lambda, solver, cg_steps,
                                            dynamic_lambda, is_x_bias_last_row, g_ws.scalars, nullptr);
  else
    rc = run_half_iteration(conf, implicit, dX.as<float>(), dY.as<float>(), dG.as<float>(), rank, lambda, solver,
                            cg_steps, dynamic_lambda, g_ws.scalars, nullptr);
  if (rc) return rc;
  double reg = 0.0;
  if (lambda > 0 && nx > 0) {  // + lambda * accu(X % X)  [* cnt_X]
    const float* Xreg = dX.as<float>();
    int kreg = rank;
    DevBuf dXe;
    if (with_biases) {  // every row of X but the ones: drop_row(X, !is_x_bias_last_row), wrmf_explicit.hpp:147-159
      kreg = rank - 1;
      HIP_TRY(dXe.alloc((size_t)kreg * n_rows * 4));
      HIP_TRY(hipMemcpy2D(dXe.p, (size_t)kreg * 4, dX.as<float>() + (is_x_bias_last_row ? 1 : 0), (size_t)rank * 4,
                          (size_t)kreg * 4, (size_t)n_rows, hipMemcpyDeviceToDevice));
      Xreg = dXe.as<float>();
    }
    hipError_t e = launch_weighted_sumsq(Xreg, kreg, n_rows, weighted ? dW.as<float>() : nullptr,
                                         g_ws.scalars + 1, g_ws.partials, nullptr);
    if (e != hipSuccess) return hip_fail(e, "launch_weighted_sumsq");
    HIP_TRY(hipDeviceSynchronize());   // dXe is released at the end of this block
  }
  HIP_TRY(hipDeviceSynchronize());
  int64_t nfail = 0;
  rsparse_hip_take_numeric_failures(&nfail, nullptr);
  double host_scalars[2] = {0, 0};
  HIP_TRY(hipMemcpy(host_scalars, g_ws.scalars, 2 * sizeof(double), hipMemcpyDeviceToHost));
  if (lambda > 0 && nx > 0) reg = lambda * host_scalars[1];
  if (ny) {
    if (sizeof(TX) == sizeof(float)) {
      HIP_TRY(hipMemcpy(Y, dY.p, ny * 4, hipMemcpyDeviceToHost));
    } else {
      tmp.resize(ny);
      HIP_TRY(hipMemcpy(tmp.data(), dY.p, ny * 4, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < ny; i++) Y[i] = (TX)tmp[i];
    }
  }
  const double nnz = (double)conf->d.nnz;
  if (loss_out) *loss_out = (host_scalars[0] + reg) / nnz;  // wrmf_implicit.hpp:304
  if (nfail)
    return fail(RSPARSE_HIP_ERR_NUMERIC, std::to_string(nfail) + " per-row systems were singular (not positive definite, and "
                                         "the general solver found a zero pivot column)");
  return RSPARSE_HIP_OK;
}

}  // namespace

namespace rsparse_hip {
int capi_fail(int code, const std::string& msg) { return fail(code, msg); }
int capi_hip_fail(hipError_t e, const char* what) { return hip_fail(e, what); }
int* capi_fail_counters() { return g_ws.ensure_device() ? nullptr : g_ws.fails; }
void capi_fail_carry_add(int64_t unresolved, int64_t fallback) {
  g_fail_carry[0] += unresolved;
  g_fail_carry[1] += fallback;
}
StaleFailures::StaleFailures() { rsparse_hip_take_numeric_failures(&unresolved, &fallback); }
void prof_note(hipEvent_t* ev_slot, const void* kernel_fn) {
  if (!ev_slot || !g_prof.on) return;
  const ptrdiff_t i = ev_slot - g_prof.ev;
  if (i >= 0 && i < 10 && !g_prof.kern[i]) g_prof.kern[i] = kernel_fn;   // the first launch of a segment names it
}
}  // namespace rsparse_hip

extern "C" {

const char* rsparse_hip_last_error(void) { return g_err.c_str(); }
int rsparse_hip_abi_version(void) { return 6; }

int rsparse_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int rsparse_hip_set_device(int device) {
  HIP_TRY(hipSetDevice(device));
  return RSPARSE_HIP_OK;
}

int rsparse_hip_csc_create_host(int n_rows, int n_cols, const int32_t* col_ptrs, const int32_t* row_indices,
                                const double* values, rsparse_hip_csc** out) {
  if (!out) return fail(RSPARSE_HIP_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (n_rows < 0 || n_cols < 0) return fail(RSPARSE_HIP_ERR_INVALID, "negative matrix dimension");
  if (!col_ptrs) return fail(RSPARSE_HIP_ERR_INVALID, "col_ptrs is NULL");
  const int64_t nnz = (int64_t)col_ptrs[n_cols] - col_ptrs[0];
  if (col_ptrs[0] != 0 || nnz < 0) return fail(RSPARSE_HIP_ERR_INVALID, "col_ptrs must start at 0 and be non-decreasing");
  if (nnz > 0 && (!row_indices || !values)) return fail(RSPARSE_HIP_ERR_INVALID, "row_indices or values is NULL");
  for (int64_t e = 0; e < nnz; e++)
    if (row_indices[e] < 0 || row_indices[e] >= n_rows)
      return fail(RSPARSE_HIP_ERR_INVALID, "row index out of range");
  rsparse_hip_csc* m = new rsparse_hip_csc();
  struct Guard { rsparse_hip_csc* c; ~Guard() { if (c) rsparse_hip_csc_destroy(c); } } guard{m};
  if (hipGetDevice(&m->device) != hipSuccess) return fail(RSPARSE_HIP_ERR_RUNTIME, "no HIP device");
  DevCSC& d = m->d;
  d.n_rows = n_rows; d.n_cols = n_cols; d.nnz = nnz; d.owns_matrix = true;
  int32_t *dp = nullptr, *di = nullptr;
  float* dv = nullptr;
  HIP_TRY(hipMalloc(&dp, ((size_t)n_cols + 1) * 4));
  d.col_ptrs = dp;
  HIP_TRY(hipMalloc(&di, (size_t)std::max<int64_t>(nnz, 4) * 4));
  d.row_idx = di;
  HIP_TRY(hipMalloc(&dv, (size_t)std::max<int64_t>(nnz, 4) * 4));
  d.vals = dv;
  HIP_TRY(hipMemcpy(dp, col_ptrs, ((size_t)n_cols + 1) * 4, hipMemcpyHostToDevice));
  if (nnz) {
    HIP_TRY(hipMemcpy(di, row_indices, (size_t)nnz * 4, hipMemcpyHostToDevice));
    // values: f64 on the wire (dgCMatrix@x), f32 once resident -- the conversion the reference does per
    // column with arma::conv_to (wrmf_implicit.hpp:182-183)
    std::vector<float> v32 = to_f32(values, (size_t)nnz);
    HIP_TRY(hipMemcpy(dv, v32.data(), (size_t)nnz * 4, hipMemcpyHostToDevice));
  }
  int rc = build_schedule(d, col_ptrs);
  if (rc) return rc;
  if ((rc = build_q_schedule(d, col_ptrs))) return rc;
  guard.c = nullptr;
  *out = m;
  return RSPARSE_HIP_OK;
}

int rsparse_hip_csc_create_device(int n_rows, int n_cols, const int32_t* d_col_ptrs, const int32_t* d_row_indices,
                                  const float* d_values, rsparse_hip_csc** out) {
  if (!out) return fail(RSPARSE_HIP_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (n_rows < 0 || n_cols < 0) return fail(RSPARSE_HIP_ERR_INVALID, "negative matrix dimension");
  if (!d_col_ptrs) return fail(RSPARSE_HIP_ERR_INVALID, "col_ptrs is NULL");
  std::vector<int32_t> hp((size_t)n_cols + 1);
  HIP_TRY(hipMemcpy(hp.data(), d_col_ptrs, hp.size() * 4, hipMemcpyDeviceToHost));
  if (hp[0] != 0) return fail(RSPARSE_HIP_ERR_INVALID, "col_ptrs must start at 0");
  const int64_t nnz = hp[n_cols];
  if (nnz > 0 && (!d_row_indices || !d_values)) return fail(RSPARSE_HIP_ERR_INVALID, "row_indices or values is NULL");
  if (nnz < 0) return fail(RSPARSE_HIP_ERR_INVALID, "col_ptrs must be non-decreasing");
  for (int c = 0; c < n_cols; c++)
    if (hp[(size_t)c + 1] < hp[(size_t)c]) return fail(RSPARSE_HIP_ERR_INVALID, "col_ptrs must be non-decreasing");
  {   // the same validation the host constructor does: the kernels gather X[row_index] unchecked
    int bad = 0;
    HIP_TRY(check_row_indices_device(d_row_indices, nnz, n_rows, nullptr, &bad));
    if (bad) return fail(RSPARSE_HIP_ERR_INVALID, "row index out of range");
  }
  rsparse_hip_csc* m = new rsparse_hip_csc();
  struct Guard { rsparse_hip_csc* c; ~Guard() { if (c) rsparse_hip_csc_destroy(c); } } guard{m};
  if (hipGetDevice(&m->device) != hipSuccess) return fail(RSPARSE_HIP_ERR_RUNTIME, "no HIP device");
  DevCSC& d = m->d;
  d.n_rows = n_rows; d.n_cols = n_cols; d.nnz = nnz; d.owns_matrix = false;
  d.col_ptrs = d_col_ptrs; d.row_idx = d_row_indices; d.vals = d_values;
  int rc = build_schedule(d, hp.data());
  if (rc) return rc;
  if ((rc = build_q_schedule(d, hp.data()))) return rc;
  guard.c = nullptr;
  *out = m;
  return RSPARSE_HIP_OK;
}

int rsparse_hip_csc_transpose_device(int n_rows, int n_cols, const int32_t* d_p, const int32_t* d_i, const float* d_x,
                                     int32_t* d_pt, int32_t* d_it, float* d_xt, void* stream) {
  if (n_rows < 0 || n_cols < 0) return fail(RSPARSE_HIP_ERR_INVALID, "negative matrix dimension");
  if (!d_p || !d_pt) return fail(RSPARSE_HIP_ERR_INVALID, "col_ptrs (input or output) is NULL");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int32_t ends[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(&ends[0], d_p, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(&ends[1], d_p + n_cols, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  const int64_t nnz = (int64_t)ends[1] - ends[0];
  if (ends[0] != 0 || nnz < 0) return fail(RSPARSE_HIP_ERR_INVALID, "col_ptrs must start at 0 and be non-decreasing");
  if (nnz > 0 && (!d_i || !d_x || !d_it || !d_xt)) return fail(RSPARSE_HIP_ERR_INVALID, "index or value array is NULL");
  int bad = 0;
  hipError_t e = transpose_csc_device(n_rows, n_cols, nnz, d_p, d_i, d_x, d_pt, d_it, d_xt, s, &bad);
  if (bad) return fail(RSPARSE_HIP_ERR_INVALID, "row index out of range");
  if (e != hipSuccess) return hip_fail(e, "transpose_csc_device");
  return RSPARSE_HIP_OK;
}

int rsparse_hip_values_to_float_device(int64_t n, const double* d_src, float* d_dst, void* stream) {
  if (n < 0) return fail(RSPARSE_HIP_ERR_INVALID, "negative length");
  if (n > 0 && (!d_src || !d_dst)) return fail(RSPARSE_HIP_ERR_INVALID, "NULL array");
  if (n == 0) return RSPARSE_HIP_OK;
  hipError_t e = launch_f64_to_f32(d_src, d_dst, (size_t)n, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return hip_fail(e, "launch_f64_to_f32");
  return RSPARSE_HIP_OK;
}

int rsparse_hip_csc_destroy(rsparse_hip_csc* m) {
  if (!m) return RSPARSE_HIP_OK;
  DevCSC& d = m->d;
  if (d.long_rows) (void)hipFree(d.long_rows);
  if (d.q_order) (void)hipFree(d.q_order);
  if (d.q_stream_off) (void)hipFree(d.q_stream_off);
  if (d.q_ne_rows) (void)hipFree(d.q_ne_rows);
  if (d.q_ne1_rows) (void)hipFree(d.q_ne1_rows);
  if (d.q_ne1_ptr) (void)hipFree(d.q_ne1_ptr);
  if (d.q_ne_ptr) (void)hipFree(d.q_ne_ptr);
  if (d.q_ne_segs) (void)hipFree(d.q_ne_segs);
  if (d.q_ne_split_rows) (void)hipFree(d.q_ne_split_rows);
  if (d.q_ne_split_ptr) (void)hipFree(d.q_ne_split_ptr);
  if (d.q_nec_own) {
    if (d.q_nec_rows) (void)hipFree(d.q_nec_rows);
    if (d.q_nec_ptr) (void)hipFree(d.q_nec_ptr);
    if (d.q_nec_segs) (void)hipFree(d.q_nec_segs);
    if (d.q_nec_split_rows) (void)hipFree(d.q_nec_split_rows);
    if (d.q_nec_split_ptr) (void)hipFree(d.q_nec_split_ptr);
  }
  if (d.vstats) (void)hipFree(d.vstats);
  if (d.owns_matrix) {
    if (d.col_ptrs) (void)hipFree(const_cast<int32_t*>(d.col_ptrs));
    if (d.row_idx) (void)hipFree(const_cast<int32_t*>(d.row_idx));
    if (d.vals) (void)hipFree(const_cast<float*>(d.vals));
  }
  delete m;
  return RSPARSE_HIP_OK;
}

int rsparse_hip_csc_freeze_values(rsparse_hip_csc* m, int frozen) {
  if (!m) return fail(RSPARSE_HIP_ERR_INVALID, "NULL handle");
  m->d.vals_frozen = frozen != 0;
  m->d.vstats_valid = false;   // (a fresh promise starts from a fresh scan)
  return RSPARSE_HIP_OK;
}

int rsparse_hip_csc_info(const rsparse_hip_csc* m, int64_t info_out[40]) {
  if (!m || !info_out) return fail(RSPARSE_HIP_ERR_INVALID, "NULL argument");
  for (int b = 0; b < 40; b++) info_out[b] = 0;
  for (int b = 0; b < 6; b++) {
    info_out[8 + b] = m->d.q_off[b + 1] - m->d.q_off[b];
    info_out[14 + b] = m->d.q_nnz[b];
  }
  info_out[20] = m->d.q_cfg;
  info_out[21] = m->d.q_ne_nseg;
  for (int b = 0; b < 6; b++) {
    info_out[22 + b] = cgq_bucket_wpr(m->d.q_cfg, b);
    info_out[28 + b] = cgq_bucket_capq(m->d.q_cfg, b);
    info_out[34 + b] = cgq_bucket_waves(m->d.q_cfg, b) * (cgq_bucket_stream(m->d.q_cfg, b) ? -1 : 1);
  }
  info_out[0] = m->d.n_rows; info_out[1] = m->d.n_cols; info_out[2] = m->d.nnz;
  info_out[3] = m->d.n_long; info_out[4] = m->d.max_len; info_out[5] = m->d.nnz_long;
  info_out[6] = m->d.n_empty; info_out[7] = m->d.short_max;
  return RSPARSE_HIP_OK;
}

int rsparse_hip_set_launch_mode(int mode) {
  if (mode < 0 || mode > 2) return fail(RSPARSE_HIP_ERR_INVALID, "launch mode must be 0, 1 or 2");
  cgq_set_launch_mode(mode);
  return RSPARSE_HIP_OK;
}

int rsparse_hip_profile_enable(int on) {
  g_prof.on = on != 0;
  g_prof.have = false;
  return RSPARSE_HIP_OK;
}

int rsparse_hip_profile_last_names(char* buf, int cap) {
  if (!buf || cap <= 0) return fail(RSPARSE_HIP_ERR_INVALID, "buf is NULL or empty");
  buf[0] = 0;
  if (!g_prof.on || !g_prof.have) return fail(RSPARSE_HIP_ERR_INVALID, "no profiled call to report");
  std::string out;
  for (int i = 0; i < g_prof.nseg; i++) {
    if (i) out += '\n';
    if (!g_prof.kern[i]) continue;
    const char* mangled = hipKernelNameRefByPtr(g_prof.kern[i], nullptr);
    if (!mangled) continue;
    int st = 0;
    char* dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &st);
    if (st != 0 || !dem) {   // this libstdc++ does not know _Float16 ("DF16_"): demangle it as half ("Dh")
      std::string alt(mangled);
      for (size_t p2; (p2 = alt.find("DF16_")) != std::string::npos;) alt.replace(p2, 5, "Dh");
      std::free(dem);
      dem = abi::__cxa_demangle(alt.c_str(), nullptr, nullptr, &st);
    }
    out += (st == 0 && dem) ? dem : mangled;
    std::free(dem);
  }
  if ((int)out.size() + 1 > cap) return fail(RSPARSE_HIP_ERR_INVALID, "buffer too small for the kernel names");
  std::memcpy(buf, out.c_str(), out.size() + 1);
  return RSPARSE_HIP_OK;
}

int rsparse_hip_profile_last(double ms_out[8]) {
  if (!ms_out) return fail(RSPARSE_HIP_ERR_INVALID, "ms_out is NULL");
  for (int i = 0; i < 8; i++) ms_out[i] = 0.0;
  if (!g_prof.on || !g_prof.have) return fail(RSPARSE_HIP_ERR_INVALID, "no profiled call to report");
  HIP_TRY(hipEventSynchronize(g_prof.ev[g_prof.nseg]));
  for (int i = 0; i < g_prof.nseg; i++) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, g_prof.ev[i], g_prof.ev[i + 1]));
    ms_out[i] = ms;
  }
  return RSPARSE_HIP_OK;
}

int rsparse_hip_gramian_device(const float* d_X, int rank, int64_t n, double lambda, float* d_XtX_out,
                               double* d_sumsq_out, void* stream) {
  return rsparse_hip_gramian_absmax_device(d_X, rank, n, lambda, d_XtX_out, d_sumsq_out, nullptr, stream);
}

int rsparse_hip_gramian_absmax_device(const float* d_X, int rank, int64_t n, double lambda, float* d_XtX_out,
                                      double* d_sumsq_out, float* d_absmax_inout, void* stream) {
  if (!d_X || !d_XtX_out) return fail(RSPARSE_HIP_ERR_INVALID, "X or XtX_out is NULL");
  if (rank <= 0 || n < 0) return fail(RSPARSE_HIP_ERR_INVALID, "rank must be positive and n non-negative");
  if (rank > RSPARSE_HIP_MAX_RANK) return fail(RSPARSE_HIP_ERR_UNSUPPORTED, "rank > 256 is not on the device path");
  int rc = g_ws.ensure_device();
  if (rc) return rc;
  if (wide_supported(rank)) {   // ranks 129..256 (d_absmax_inout is left as it is: only the rank <= 128 long-row kernel reads it)
    if ((rc = g_ws.ensure_gram(wide_gramian_scratch_floats(rank)))) return rc;
    hipError_t we = launch_gramian_wide(d_X, rank, n, (float)lambda, d_XtX_out, d_sumsq_out, g_ws.gram, (hipStream_t)stream);
    if (we != hipSuccess) return hip_fail(we, "launch_gramian_wide");
    return RSPARSE_HIP_OK;
  }
  if ((rc = g_ws.ensure_gram(gramian_scratch_floats(rank, n)))) return rc;
  const float ridge = (float)lambda;  // float::fl(diag(lambda)), R/model_WRMF.R:476
  hipEvent_t* ev = g_prof.begin();
  hipError_t e = launch_gramian(d_X, rank, n, ridge, d_XtX_out, d_sumsq_out, g_ws.gram, (hipStream_t)stream, ev,
                                reinterpret_cast<unsigned*>(d_absmax_inout));
  if (e != hipSuccess) return hip_fail(e, "launch_gramian");
  if (ev) {
    g_prof.have = true;
    g_prof.nseg = 2;
  }
  return RSPARSE_HIP_OK;
}

int rsparse_hip_gramian_float(const float* X, int rank, int64_t n, double lambda, float* XtX_out) {
  if (!X || !XtX_out) return fail(RSPARSE_HIP_ERR_INVALID, "X or XtX_out is NULL");
  if (rank <= 0 || n < 0) return fail(RSPARSE_HIP_ERR_INVALID, "rank must be positive and n non-negative");
  DevBuf dX, dG;
  HIP_TRY(dX.alloc((size_t)rank * n * 4));
  HIP_TRY(dG.alloc((size_t)rank * rank * 4));
  if (n) HIP_TRY(hipMemcpy(dX.p, X, (size_t)rank * n * 4, hipMemcpyHostToDevice));
  int rc = rsparse_hip_gramian_device(dX.as<float>(), rank, n, lambda, dG.as<float>(), nullptr, nullptr);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(XtX_out, dG.p, (size_t)rank * rank * 4, hipMemcpyDeviceToHost));
  return RSPARSE_HIP_OK;
}

int rsparse_hip_als_implicit_device(const rsparse_hip_csc* conf, const float* d_X, float* d_Y, const float* d_XtX,
                                    int rank, double lambda, unsigned solver, unsigned cg_steps,
                                    const float* d_absmax, double* d_loss_rows_out, void* stream) {
  return run_half_iteration(conf, true, d_X, d_Y, d_XtX, rank, lambda, solver, cg_steps, 0, d_loss_rows_out,
                            (hipStream_t)stream, nullptr, d_absmax);
}

int rsparse_hip_als_explicit_device(const rsparse_hip_csc* conf, const float* d_X, float* d_Y, int rank,
                                    double lambda, unsigned solver, unsigned cg_steps, int dynamic_lambda,
                                    double* d_loss_rows_out, void* stream) {
  return run_half_iteration(conf, false, d_X, d_Y, nullptr, rank, lambda, solver, cg_steps, dynamic_lambda,
                            d_loss_rows_out, (hipStream_t)stream);
}

int rsparse_hip_als_implicit_bias_device(const rsparse_hip_csc* conf, const float* d_X, float* d_Y, const float* d_XtX,
                                         int rank, double lambda, unsigned solver, int is_x_bias_last_row,
                                         double* d_loss_rows_out, void* stream) {
  int rc = check_variant(solver, 1, 0.0, true);
  if (rc) return rc;
  return run_half_iteration_implicit_biased(conf, d_X, d_Y, d_XtX, rank, lambda, solver, is_x_bias_last_row,
                                            d_loss_rows_out, (hipStream_t)stream);
}

int rsparse_hip_als_implicit_global_bias_device(const rsparse_hip_csc* conf, const float* d_X, float* d_Y,
                                                const float* d_XtX, int rank, double lambda, unsigned solver,
                                                unsigned cg_steps, int with_biases, int is_x_bias_last_row,
                                                double global_bias, int double_threshold, const float* d_absmax,
                                                double* d_loss_rows_out, void* stream) {
  int rc = check_variant(solver, with_biases, global_bias, true);
  if (rc) return rc;
  if (with_biases)
    return run_half_iteration_implicit_biased(conf, d_X, d_Y, d_XtX, rank, lambda, solver, is_x_bias_last_row,
                                              d_loss_rows_out, (hipStream_t)stream, global_bias, double_threshold != 0);
  if (!has_global_bias(global_bias, double_threshold != 0))
    return run_half_iteration(conf, true, d_X, d_Y, d_XtX, rank, lambda, solver, cg_steps, 0, d_loss_rows_out,
                              (hipStream_t)stream, nullptr, d_absmax);
  return run_half_iteration_implicit_global(conf, d_X, d_Y, d_XtX, rank, lambda, solver, cg_steps, global_bias, nullptr,
                                            nullptr, d_loss_rows_out, (hipStream_t)stream, d_absmax);
}

int rsparse_hip_initialize_biases_implicit_device(const rsparse_hip_csc* c_ui, const rsparse_hip_csc* c_iu,
                                                  float* d_user_bias, float* d_item_bias, double lambda,
                                                  int non_negative, int calculate_global_bias,
                                                  double* global_bias_out, void* stream) {
  if (!c_ui || !c_iu || !d_user_bias || !d_item_bias) return fail(RSPARSE_HIP_ERR_INVALID, "NULL matrix or bias vector");
  const DevCSC& a = c_ui->d;   // users x items, columns = items
  const DevCSC& b = c_iu->d;   // items x users, columns = users
  if (a.n_rows != b.n_cols || a.n_cols != b.n_rows || a.nnz != b.nnz)
    return fail(RSPARSE_HIP_ERR_INVALID, "the two matrices are not transposes of each other");
  hipStream_t s = (hipStream_t)stream;
  int rc = g_ws.ensure_device();
  if (rc) return rc;
  if ((rc = g_ws.ensure_partials(1024))) return rc;
  const int n_items = a.n_cols, n_users = b.n_cols;
  DevBuf stats;   // means / adjustments of both sides (wrmf_utils.hpp:97-124), doubles
  HIP_TRY(stats.alloc(((size_t)2 * n_items + (size_t)2 * n_users + 4) * sizeof(double)));
  double* item_means = stats.as<double>();
  double* item_adj = item_means + n_items;
  double* user_means = item_adj + n_items;
  double* user_adj = user_means + n_users;
  hipError_t e;
  if ((e = launch_bias_implicit_prep(a.col_ptrs, a.vals, n_items, n_users, lambda, item_means, item_adj, s)) != hipSuccess ||
      (e = launch_bias_implicit_prep(b.col_ptrs, b.vals, n_users, n_items, lambda, user_means, user_adj, s)) != hipSuccess)
    return hip_fail(e, "launch_bias_implicit_prep");
  double global_bias = 0.0;
  if (calculate_global_bias) {   // :90-93: sum(x) / (sum(x) + n_users n_items - nnz)
    if ((e = launch_values_sum(a.vals, a.nnz, g_ws.partials, g_ws.scalars + 2, s)) != hipSuccess)
      return hip_fail(e, "launch_values_sum");
    double sum = 0.0;
    HIP_TRY(hipMemcpyAsync(&sum, g_ws.scalars + 2, sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    global_bias = sum / (sum + (double)n_users * (double)n_items - (double)a.nnz);
  }
  if (non_negative) global_bias = std::fmax(0.0, global_bias);
  if (global_bias_out) *global_bias_out = global_bias;
  for (int iter = 0; iter < 5; iter++) {   // :130-162
    const double* usum = nullptr;
    if (iter > 0) {                        // mean of the user biases of the previous sweep (:131-135)
      if ((e = launch_values_sum(d_user_bias, n_users, g_ws.partials, g_ws.scalars + 2, s)) != hipSuccess)
        return hip_fail(e, "launch_values_sum");
      usum = g_ws.scalars + 2;
    }
    if ((e = launch_bias_implicit_sweep(a.col_ptrs, a.row_idx, a.vals, d_user_bias, n_items, n_users, usum, item_means,
                                        item_adj, non_negative, global_bias, d_item_bias, s)) != hipSuccess)
      return hip_fail(e, "launch_bias_implicit_sweep");
    if ((e = launch_values_sum(d_item_bias, n_items, g_ws.partials, g_ws.scalars + 3, s)) != hipSuccess)
      return hip_fail(e, "launch_values_sum");
    if ((e = launch_bias_implicit_sweep(b.col_ptrs, b.row_idx, b.vals, d_item_bias, n_users, n_items, g_ws.scalars + 3,
                                        user_means, user_adj, non_negative, global_bias, d_user_bias, s)) != hipSuccess)
      return hip_fail(e, "launch_bias_implicit_sweep");
  }
  HIP_TRY(hipStreamSynchronize(s));   // `stats` is released on return
  return RSPARSE_HIP_OK;
}

// ---- single sweeps of the bias initialisation over ONE column block (sharded drivers; see the header) ----
int rsparse_hip_bias_sweep_explicit_device(const rsparse_hip_csc* conf, const float* d_other_bias, double lambda,
                                           int dynamic_lambda, int non_negative, float* d_out, void* stream) {
  if (!conf || !d_other_bias || !d_out) return fail(RSPARSE_HIP_ERR_INVALID, "NULL matrix or bias vector");
  const DevCSC& a = conf->d;
  hipError_t e = launch_bias_sweep(a.col_ptrs, a.row_idx, a.vals, d_other_bias, a.n_cols, (float)lambda, dynamic_lambda,
                                   non_negative, d_out, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "launch_bias_sweep");
  return RSPARSE_HIP_OK;
}

int rsparse_hip_bias_prep_implicit_device(const rsparse_hip_csc* conf, int n_other, double lambda, double* d_means,
                                          double* d_adj, void* stream) {
  if (!conf || !d_means || !d_adj) return fail(RSPARSE_HIP_ERR_INVALID, "NULL matrix or output");
  const DevCSC& a = conf->d;
  hipError_t e = launch_bias_implicit_prep(a.col_ptrs, a.vals, a.n_cols, n_other, lambda, d_means, d_adj, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "launch_bias_implicit_prep");
  return RSPARSE_HIP_OK;
}

int rsparse_hip_bias_sweep_implicit_device(const rsparse_hip_csc* conf, const float* d_other_bias, int n_other,
                                           const double* d_other_sum, const double* d_means, const double* d_adj,
                                           int non_negative, double global_bias, float* d_out, void* stream) {
  if (!conf || !d_other_bias || !d_means || !d_adj || !d_out) return fail(RSPARSE_HIP_ERR_INVALID, "NULL matrix or vector");
  const DevCSC& a = conf->d;
  hipError_t e = launch_bias_implicit_sweep(a.col_ptrs, a.row_idx, a.vals, d_other_bias, a.n_cols, n_other, d_other_sum, d_means,
                                            d_adj, non_negative, global_bias, d_out, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "launch_bias_implicit_sweep");
  return RSPARSE_HIP_OK;
}

int rsparse_hip_als_explicit_bias_device(const rsparse_hip_csc* conf, const float* d_X, float* d_Y, int rank,
                                         double lambda, unsigned solver, unsigned cg_steps, int dynamic_lambda,
                                         int is_x_bias_last_row, double* d_loss_rows_out, void* stream) {
  int rc = check_variant(solver, 1, 0.0, false);
  if (rc) return rc;
  return run_half_iteration_explicit_biased(conf, d_X, d_Y, rank, lambda, solver, cg_steps, dynamic_lambda,
                                            is_x_bias_last_row, d_loss_rows_out, (hipStream_t)stream);
}

int rsparse_hip_initialize_biases_explicit_device(rsparse_hip_csc* c_ui, rsparse_hip_csc* c_iu, float* d_user_bias,
                                                  float* d_item_bias, double lambda, int dynamic_lambda,
                                                  int non_negative, int calculate_global_bias,
                                                  double* global_bias_out, void* stream) {
  if (!c_ui || !c_iu || !d_user_bias || !d_item_bias) return fail(RSPARSE_HIP_ERR_INVALID, "NULL matrix or bias vector");
  const DevCSC& a = c_ui->d;   // users x items, columns = items
  const DevCSC& b = c_iu->d;   // items x users, columns = users
  if (a.n_rows != b.n_cols || a.n_cols != b.n_rows || a.nnz != b.nnz)
    return fail(RSPARSE_HIP_ERR_INVALID, "the two matrices are not transposes of each other");
  hipStream_t s = (hipStream_t)stream;
  int rc = g_ws.ensure_device();
  if (rc) return rc;
  if ((rc = g_ws.ensure_partials(1024))) return rc;
  double global_bias = 0.0;
  hipError_t e;
  if (calculate_global_bias && a.nnz > 0) {   // wrmf_utils.hpp:41-52: mean of the values, removed from both orientations
    if (a.vals_frozen || b.vals_frozen)
      return fail(RSPARSE_HIP_ERR_INVALID, "the values of a frozen handle cannot lose their mean (rsparse_hip_csc_freeze_values)");
    if ((e = launch_values_sum(a.vals, a.nnz, g_ws.partials, g_ws.scalars + 2, s)) != hipSuccess)
      return hip_fail(e, "launch_values_sum");
    const double inv = 1.0 / (double)a.nnz;
    if ((e = launch_values_subtract_mean(const_cast<float*>(a.vals), a.nnz, g_ws.scalars + 2, inv, s)) != hipSuccess ||
        (e = launch_values_subtract_mean(const_cast<float*>(b.vals), b.nnz, g_ws.scalars + 2, inv, s)) != hipSuccess)
      return hip_fail(e, "launch_values_subtract_mean");
    double sum = 0.0;
    HIP_TRY(hipMemcpyAsync(&sum, g_ws.scalars + 2, sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    global_bias = sum * inv;
  }
  for (int iter = 0; iter < 5; iter++) {       // :54-82
    if ((e = launch_bias_sweep(a.col_ptrs, a.row_idx, a.vals, d_user_bias, a.n_cols, (float)lambda, dynamic_lambda,
                               non_negative, d_item_bias, s)) != hipSuccess ||
        (e = launch_bias_sweep(b.col_ptrs, b.row_idx, b.vals, d_item_bias, b.n_cols, (float)lambda, dynamic_lambda,
                               non_negative, d_user_bias, s)) != hipSuccess)
      return hip_fail(e, "launch_bias_sweep");
  }
  if (global_bias_out) *global_bias_out = global_bias;
  return RSPARSE_HIP_OK;
}

int rsparse_hip_values_subtract_mean_device(int64_t n, float* d_x, float* d_x_other, double* mean_out, void* stream) {
  if (n < 0) return fail(RSPARSE_HIP_ERR_INVALID, "negative length");
  if (mean_out) *mean_out = 0.0;
  if (n == 0) return RSPARSE_HIP_OK;
  if (!d_x) return fail(RSPARSE_HIP_ERR_INVALID, "values is NULL");
  hipStream_t s = (hipStream_t)stream;
  int rc = g_ws.ensure_device();
  if (rc) return rc;
  if ((rc = g_ws.ensure_partials(1024))) return rc;
  hipError_t e = launch_values_sum(d_x, n, g_ws.partials, g_ws.scalars + 2, s);
  if (e != hipSuccess) return hip_fail(e, "launch_values_sum");
  const double inv = 1.0 / (double)n;
  if ((e = launch_values_subtract_mean(d_x, n, g_ws.scalars + 2, inv, s)) != hipSuccess)
    return hip_fail(e, "launch_values_subtract_mean");
  if (d_x_other && (e = launch_values_subtract_mean(d_x_other, n, g_ws.scalars + 2, inv, s)) != hipSuccess)
    return hip_fail(e, "launch_values_subtract_mean");
  double sum = 0.0;
  HIP_TRY(hipMemcpyAsync(&sum, g_ws.scalars + 2, sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (mean_out) *mean_out = sum * inv;
  return RSPARSE_HIP_OK;
}

int rsparse_hip_weighted_sumsq_device(const float* d_X, int rank, int64_t n, const float* d_w, double* d_out,
                                      void* stream) {
  if (!d_X || !d_out) return fail(RSPARSE_HIP_ERR_INVALID, "X or out is NULL");
  if (rank <= 0 || n < 0) return fail(RSPARSE_HIP_ERR_INVALID, "rank must be positive and n non-negative");
  int rc = g_ws.ensure_device();
  if (rc) return rc;
  if ((rc = g_ws.ensure_partials(1024))) return rc;
  hipError_t e = launch_weighted_sumsq(d_X, rank, n, d_w, d_out, g_ws.partials, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "launch_weighted_sumsq");
  return RSPARSE_HIP_OK;
}

int rsparse_hip_top_product_device(const float* d_U, const float* d_V, int n_users, int n_items, int rank, int k,
                                   const int32_t* d_nr_p, const int32_t* d_nr_j, const int32_t* d_excl0,
                                   int n_exclude, double glob_mean, int32_t* d_res, float* d_scores, void* stream) {
  if (!d_U || !d_V || !d_res || !d_scores) return fail(RSPARSE_HIP_ERR_INVALID, "NULL matrix or output");
  if (n_users < 0 || n_items < 0 || rank <= 0 || k < 1) return fail(RSPARSE_HIP_ERR_INVALID, "bad dimensions");
  if (rank > RSPARSE_HIP_MAX_RANK) return fail(RSPARSE_HIP_ERR_UNSUPPORTED, "rank > 256 is not on the device path");
  if (k > RSPARSE_HIP_MAX_TOPK) return fail(RSPARSE_HIP_ERR_UNSUPPORTED, "k > 256 is not on the device path");
  if (n_exclude > 0 && !d_excl0) return fail(RSPARSE_HIP_ERR_INVALID, "exclude is NULL");
  // (few users over many items: the items are split over the workgroups, the slices' lists land in this scratch)
  float* scratch = nullptr;
  const size_t sfl = top_product_scratch_floats(n_users, n_items, rank, k);
  if (sfl > 0) {
    int rc = g_ws.ensure_pad(sfl);
    if (rc) return rc;
    scratch = g_ws.pad_buf;
  }
  hipError_t e = launch_top_product(d_U, d_V, n_users, n_items, rank, k, d_nr_p, d_nr_p ? d_nr_j : nullptr, d_excl0,
                                    n_exclude, (float)glob_mean, d_res, d_scores, (hipStream_t)stream, scratch);
  if (e != hipSuccess) return hip_fail(e, "launch_top_product");
  return RSPARSE_HIP_OK;
}

int rsparse_hip_top_product_f64_device(const float* d_U, const float* d_V, const double* d_U64, const double* d_V64,
                                       int n_users, int n_items, int rank, int k, int extra, const int32_t* d_nr_p,
                                       const int32_t* d_nr_j, const int32_t* d_excl0, int n_exclude, double glob_mean,
                                       int32_t* d_res, double* d_scores, void* stream) {
  if (!d_U || !d_V || !d_res || !d_scores) return fail(RSPARSE_HIP_ERR_INVALID, "NULL matrix or output");
  if ((d_U64 == nullptr) != (d_V64 == nullptr)) return fail(RSPARSE_HIP_ERR_INVALID, "the double factors come as a pair");
  if (n_users < 0 || n_items < 0 || rank <= 0 || k < 1) return fail(RSPARSE_HIP_ERR_INVALID, "bad dimensions");
  if (rank > RSPARSE_HIP_MAX_RANK) return fail(RSPARSE_HIP_ERR_UNSUPPORTED, "rank > 256 is not on the device path");
  if (k > RSPARSE_HIP_MAX_TOPK) return fail(RSPARSE_HIP_ERR_UNSUPPORTED, "k > 256 is not on the device path");
  if (n_users == 0) return RSPARSE_HIP_OK;
  // candidates per user: k + extra (default: a quarter of k, at least 8), never more than the kernel's 256 or the items
  if (extra < 0) extra = std::max(8, k / 4);
  const int kc = std::max(k, std::min(std::min(k + extra, RSPARSE_HIP_MAX_TOPK), std::max(n_items, 1)));
  int rc = g_ws.ensure_bias(top_product_f64_scratch_words(n_users, kc, k));   // (nothing else uses this buffer meanwhile)
  if (rc) return rc;
  float* split = nullptr;   // few users over many items: the nominating pass splits the items over the workgroups
  const size_t sfl = top_product_scratch_floats(n_users, n_items, rank, kc);   // (or the candidate buffers of a large k)
  if (sfl > 0) {
    if ((rc = g_ws.ensure_pad(sfl))) return rc;
    split = g_ws.pad_buf;
  }
  hipError_t e = launch_top_product_f64(d_U, d_V, d_U64, d_V64, n_users, n_items, rank, k, kc, d_nr_p, d_nr_p ? d_nr_j : nullptr,
                                        d_excl0, n_exclude, glob_mean, d_res, d_scores, (hipStream_t)stream, g_ws.bias_buf, split);
  if (e != hipSuccess) return hip_fail(e, "launch_top_product_f64");
  return RSPARSE_HIP_OK;
}

int rsparse_hip_top_product(const double* x, const double* y, int nr, int nc, int rank, unsigned k, unsigned n_threads,
                            const int32_t* nr_p, const int32_t* nr_j, const int32_t* exclude, int n_exclude,
                            double glob_mean, int32_t* res, double* scores) {
  (void)n_threads;
  if (!x || !y || !res || !scores) return fail(RSPARSE_HIP_ERR_INVALID, "NULL matrix or output");
  if (nr < 0 || nc < 0 || rank <= 0 || k < 1) return fail(RSPARSE_HIP_ERR_INVALID, "bad dimensions");
  if (rank > RSPARSE_HIP_MAX_RANK) return fail(RSPARSE_HIP_ERR_UNSUPPORTED, "rank > 256 is not on the device path");
  if (k > RSPARSE_HIP_MAX_TOPK) return fail(RSPARSE_HIP_ERR_UNSUPPORTED, "k > 256 is not on the device path");
  if (n_exclude < 0 || (n_exclude > 0 && !exclude)) return fail(RSPARSE_HIP_ERR_INVALID, "bad exclude");
  // x is nr x rank column-major -> row-major fp32; y (rank x nc column-major) already has item vectors contiguous
  std::vector<float> U((size_t)nr * rank), V((size_t)nc * rank);
  std::vector<double> U64((size_t)nr * rank);
  for (int j = 0; j < nr; j++)
    for (int r = 0; r < rank; r++) {
      U64[(size_t)j * rank + r] = x[(size_t)r * nr + j];
      U[(size_t)j * rank + r] = (float)x[(size_t)r * nr + j];
    }
  for (size_t e = 0; e < V.size(); e++) V[e] = (float)y[e];
  std::vector<int32_t> ex;
  for (int e = 0; e < n_exclude; e++)
    if (exclude[e] >= 1 && exclude[e] <= nc) ex.push_back(exclude[e] - 1);   // R indices are 1-based
  std::sort(ex.begin(), ex.end());
  ex.erase(std::unique(ex.begin(), ex.end()), ex.end());
  const int64_t nr_nnz = (nr_p && nr > 0) ? (int64_t)nr_p[nr] : 0;
  DevBuf dU, dV, dU64, dV64, dP, dJ, dE, dR, dS;
  HIP_TRY(dU.alloc(U.size() * 4));
  HIP_TRY(dV.alloc(V.size() * 4));
  HIP_TRY(dU64.alloc(U64.size() * 8));
  HIP_TRY(dV64.alloc(V.size() * 8));
  HIP_TRY(dR.alloc((size_t)nr * k * 4));
  HIP_TRY(dS.alloc((size_t)nr * k * 8));
  if (!U.empty()) HIP_TRY(hipMemcpy(dU.p, U.data(), U.size() * 4, hipMemcpyHostToDevice));
  if (!V.empty()) HIP_TRY(hipMemcpy(dV.p, V.data(), V.size() * 4, hipMemcpyHostToDevice));
  if (!U64.empty()) HIP_TRY(hipMemcpy(dU64.p, U64.data(), U64.size() * 8, hipMemcpyHostToDevice));
  if (!V.empty()) HIP_TRY(hipMemcpy(dV64.p, y, V.size() * 8, hipMemcpyHostToDevice));   // rank x nc column-major = item vectors contiguous
  const bool filter = nr_nnz > 0 && nr_j;   // `not_empty_filter_matrix`, matrix_top_product.cpp:33
  if (filter) {
    HIP_TRY(dP.alloc(((size_t)nr + 1) * 4));
    HIP_TRY(dJ.alloc((size_t)nr_nnz * 4));
    HIP_TRY(hipMemcpy(dP.p, nr_p, ((size_t)nr + 1) * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dJ.p, nr_j, (size_t)nr_nnz * 4, hipMemcpyHostToDevice));
  }
  if (!ex.empty()) {
    HIP_TRY(dE.alloc(ex.size() * 4));
    HIP_TRY(hipMemcpy(dE.p, ex.data(), ex.size() * 4, hipMemcpyHostToDevice));
  }
  // candidates from the fp32 pass, scores and order from the doubles as given (find_top_product multiplies arma::mat)
  int rc = rsparse_hip_top_product_f64_device(dU.as<float>(), dV.as<float>(), dU64.as<double>(), dV64.as<double>(), nr, nc,
                                              rank, (int)k, -1, filter ? dP.as<int32_t>() : nullptr,
                                              filter ? dJ.as<int32_t>() : nullptr, ex.empty() ? nullptr : dE.as<int32_t>(),
                                              (int)ex.size(), glob_mean, dR.as<int32_t>(), dS.as<double>(), nullptr);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  std::vector<int32_t> hr((size_t)nr * k);
  std::vector<double> hs((size_t)nr * k);
  if (!hr.empty()) {
    HIP_TRY(hipMemcpy(hr.data(), dR.p, hr.size() * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hs.data(), dS.p, hs.size() * 8, hipMemcpyDeviceToHost));
  }
  for (int j = 0; j < nr; j++)
    for (unsigned c = 0; c < k; c++) {
      res[(size_t)c * nr + j] = hr[(size_t)j * k + c];
      scores[(size_t)c * nr + j] = hs[(size_t)j * k + c];
    }
  return RSPARSE_HIP_OK;
}

int rsparse_hip_take_numeric_failures(int64_t* unresolved_out, int64_t* fallback_out) {
  if (!unresolved_out) return fail(RSPARSE_HIP_ERR_INVALID, "unresolved_out is NULL");
  // counts a stateless call found on the device when it started and set aside (see StaleFailures): still the resident layer's
  *unresolved_out = g_fail_carry[0];
  if (fallback_out) *fallback_out = g_fail_carry[1];
  g_fail_carry[0] = g_fail_carry[1] = 0;
  if (!g_ws.fails) return RSPARSE_HIP_OK;
  int v[4] = {0, 0, 0, 0};
  HIP_TRY(hipMemcpy(v, g_ws.fails, sizeof(v), hipMemcpyDeviceToHost));   // (synchronises with the device)
  if (v[0] | v[1] | v[2] | v[3]) HIP_TRY(hipMemset(g_ws.fails, 0, sizeof(v)));
  // rows beyond the list's capacity could not be handed over: they count as unresolved, not as re-solved
  const int64_t lost = v[0] > kFailCap ? (int64_t)v[0] - kFailCap : 0;
  const int64_t sent = (int64_t)std::min(v[0], kFailCap) + v[2];
  *unresolved_out += (int64_t)v[1] + v[3] + lost;
  if (fallback_out) *fallback_out += sent;
  return RSPARSE_HIP_OK;
}

int rsparse_hip_als_implicit_float(int n_rows, int n_cols, const int32_t* col_ptrs, const int32_t* row_indices,
                                   const double* values, const float* X, float* Y, const float* XtX, int rank,
                                   double lambda, int n_threads, unsigned solver, unsigned cg_steps,
                                   int with_biases, int is_x_bias_last_row, double global_bias,
                                   float* global_bias_base, int global_bias_base_len, int initialize_bias_base,
                                   double* loss_out) {
  (void)n_threads;
  int rc = check_common(n_rows, n_cols, col_ptrs, row_indices, values, X, Y, rank);
  if (rc) return rc;
  if ((rc = check_variant(solver, with_biases, global_bias))) return rc;
  return stateless<float>(true, n_rows, n_cols, col_ptrs, row_indices, values, X, Y, XtX, nullptr, rank, lambda,
                          solver, cg_steps, 0, loss_out, with_biases, is_x_bias_last_row, global_bias, global_bias_base,
                          global_bias_base_len, initialize_bias_base);
}

}  // extern "C"

namespace {

// Stateless counterpart of the .Call targets _rsparse_initialize_biases_{double,float} (src/wrmf_init.cpp:5-34,
// src/RcppExports.cpp:417-454): the two S4 matrices flattened to their slots, bias vectors in TX.
template <class TX>
int initialize_biases_host(int n_users, int n_items, const int32_t* csc_p, const int32_t* csc_i, double* csc_x,
                           const int32_t* csr_p, const int32_t* csr_i, double* csr_x, TX* user_bias, TX* item_bias,
                           double lambda, int dynamic_lambda, int non_negative, int calculate_global_bias,
                           int is_explicit_feedback, double* global_bias_out) {
  if (!user_bias || !item_bias) return fail(RSPARSE_HIP_ERR_INVALID, "user_bias or item_bias is NULL");
  rsparse_hip_csc *c_ui = nullptr, *c_iu = nullptr;
  int rc = rsparse_hip_csc_create_host(n_users, n_items, csc_p, csc_i, csc_x, &c_ui);   // columns = items
  if (rc) return rc;
  struct Guard { rsparse_hip_csc* c; ~Guard() { rsparse_hip_csc_destroy(c); } } g1{c_ui};
  if ((rc = rsparse_hip_csc_create_host(n_items, n_users, csr_p, csr_i, csr_x, &c_iu))) return rc;   // columns = users
  Guard g2{c_iu};
  DevBuf dU, dI;
  HIP_TRY(dU.alloc((size_t)n_users * 4));
  HIP_TRY(dI.alloc((size_t)n_items * 4));
  std::vector<float> tmp = to_f32(user_bias, (size_t)n_users);
  if (n_users) HIP_TRY(hipMemcpy(dU.p, tmp.data(), (size_t)n_users * 4, hipMemcpyHostToDevice));
  tmp = to_f32(item_bias, (size_t)n_items);
  if (n_items) HIP_TRY(hipMemcpy(dI.p, tmp.data(), (size_t)n_items * 4, hipMemcpyHostToDevice));
  double gb = 0.0;
  if (is_explicit_feedback)
    rc = rsparse_hip_initialize_biases_explicit_device(c_ui, c_iu, dU.as<float>(), dI.as<float>(), lambda, dynamic_lambda,
                                                       non_negative, calculate_global_bias, &gb, nullptr);
  else
    rc = rsparse_hip_initialize_biases_implicit_device(c_ui, c_iu, dU.as<float>(), dI.as<float>(), lambda, non_negative,
                                                       calculate_global_bias, &gb, nullptr);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  tmp.resize((size_t)std::max(n_users, n_items));
  if (n_users) HIP_TRY(hipMemcpy(tmp.data(), dU.p, (size_t)n_users * 4, hipMemcpyDeviceToHost));
  for (int e = 0; e < n_users; e++) user_bias[e] = (TX)tmp[(size_t)e];
  if (n_items) HIP_TRY(hipMemcpy(tmp.data(), dI.p, (size_t)n_items * 4, hipMemcpyDeviceToHost));
  for (int e = 0; e < n_items; e++) item_bias[e] = (TX)tmp[(size_t)e];
  if (is_explicit_feedback && calculate_global_bias) {
    // the reference removes the global mean from the @x slots of BOTH matrices in place (wrmf_utils.hpp:41-52)
    const int64_t nnz = csc_p[n_items];
    for (int64_t e = 0; e < nnz; e++) csc_x[e] -= gb;
    for (int64_t e = 0; e < nnz; e++) csr_x[e] -= gb;
  }
  if (global_bias_out) *global_bias_out = gb;
  return RSPARSE_HIP_OK;
}

}  // namespace

extern "C" {

int rsparse_hip_initialize_biases_float(int n_users, int n_items, const int32_t* csc_p, const int32_t* csc_i,
                                        double* csc_x, const int32_t* csr_p, const int32_t* csr_i, double* csr_x,
                                        float* user_bias, float* item_bias, double lambda, int dynamic_lambda,
                                        int non_negative, int calculate_global_bias, int is_explicit_feedback,
                                        double* global_bias_out) {
  return initialize_biases_host<float>(n_users, n_items, csc_p, csc_i, csc_x, csr_p, csr_i, csr_x, user_bias, item_bias,
                                       lambda, dynamic_lambda, non_negative, calculate_global_bias, is_explicit_feedback,
                                       global_bias_out);
}
}  // extern "C"

namespace {
}  // namespace

extern "C" {

int rsparse_hip_als_explicit_float(int n_rows, int n_cols, const int32_t* col_ptrs, const int32_t* row_indices,
                                   const double* values, const float* X, float* Y, const float* cnt_X, int rank,
                                   double lambda, unsigned n_threads, unsigned solver, unsigned cg_steps,
                                   int dynamic_lambda, int with_biases, int is_x_bias_last_row, double* loss_out) {
  (void)n_threads;
  int rc = check_common(n_rows, n_cols, col_ptrs, row_indices, values, X, Y, rank);
  if (rc) return rc;
  if ((rc = check_variant(solver, with_biases, 0.0, false))) return rc;
  return stateless<float>(false, n_rows, n_cols, col_ptrs, row_indices, values, X, Y, nullptr, cnt_X, rank, lambda,
                          solver, cg_steps, dynamic_lambda, loss_out, with_biases, is_x_bias_last_row);
}

}  // extern "C"
